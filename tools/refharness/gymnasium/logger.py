import warnings


def warn(msg, *args):
    warnings.warn(msg % args if args else msg)
