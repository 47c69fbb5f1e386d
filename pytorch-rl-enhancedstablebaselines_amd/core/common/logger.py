"""Minimal logger with the reference's surface (core/common/logger.py:477-545): record / record_mean / dump /
name_to_value, stdout table when verbose. Values may be LAZY device scalars (`DeviceMean`): they are only
read back (one host sync) when the table is dumped, so the train loop never calls .item()."""
import sys
import time
from collections import defaultdict
from typing import Any, Optional


class DeviceMean:
    """Mean of `count` device-side loss values kept as one device tensor; float() syncs."""

    def __init__(self, total, count: int):
        self.total, self.count = total, count

    def __float__(self):
        return float(self.total) / max(self.count, 1)

    def __repr__(self):
        return f"{float(self):.6g}"


class Logger:
    def __init__(self, folder: Optional[str] = None, output_formats=None, verbose: int = 0):
        self.name_to_value: dict = defaultdict(float)
        self.name_to_count: dict = defaultdict(int)
        self.name_to_excluded: dict = {}
        self.dir = folder
        self.output_formats = list(output_formats or [])
        self.verbose = verbose
        self.dump_count = 0

    def record(self, key: str, value: Any, exclude=None) -> None:
        self.name_to_value[key] = value
        self.name_to_excluded[key] = exclude

    def record_mean(self, key: str, value, exclude=None) -> None:
        if value is None:
            return
        old, count = self.name_to_value[key], self.name_to_count[key]
        self.name_to_value[key] = old * count / (count + 1) + value / (count + 1)
        self.name_to_count[key] = count + 1
        self.name_to_excluded[key] = exclude

    def resolved(self) -> dict:
        return {k: (float(v) if isinstance(v, DeviceMean) else v) for k, v in self.name_to_value.items()}

    def dump(self, step: int = 0) -> None:
        vals = self.resolved()
        self.last_dump = dict(vals, step=step)
        self.dump_count += 1
        for fmt in self.output_formats:
            fmt.write(vals, self.name_to_excluded, step)
        if self.verbose >= 1:
            width = max((len(k) for k in vals), default=10)
            out = ["-" * (width + 18)]
            for k in sorted(vals):
                v = vals[k]
                out.append(f"| {k:<{width}} | {v:<12.5g} |" if isinstance(v, float) else f"| {k:<{width}} | {str(v):<12} |")
            out.append("-" * (width + 18))
            sys.stdout.write("\n".join(out) + "\n")
        self.name_to_value.clear()
        self.name_to_count.clear()
        self.name_to_excluded.clear()

    def get_dir(self):
        return self.dir

    def close(self):
        for fmt in self.output_formats:
            if hasattr(fmt, "close"):
                fmt.close()


def configure_logger(verbose: int = 0, tensorboard_log=None, tb_log_name: str = "", reset_num_timesteps: bool = True) -> Logger:
    """reference: core/common/utils.py:189-223 (stdout only; CSV/TensorBoard writers are out of scope)"""
    return Logger(folder=None, output_formats=[], verbose=verbose)


def now_ns() -> int:
    return time.time_ns()
