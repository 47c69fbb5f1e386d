class Error(Exception):
    pass


class DependencyNotInstalled(Error):
    pass
