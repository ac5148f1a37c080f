"""Console helpers with the reference's names (main_code/utils/utils.py:3-56)."""
import sys


class Tee:
    """Fan writes out to several text streams (entry scripts tee stdout into a log file)."""

    def __init__(self, *streams):
        self.files = streams
        self.primary = streams[0] if streams else sys.stdout

    def write(self, text):
        for f in self.files:
            f.write(text)

    def flush(self):
        for f in self.files:
            f.flush()

    def fileno(self):
        return self.primary.fileno()


class AverageMeter:
    """Running weighted mean: update(value, n) adds n samples worth `value` each."""

    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        spec = self.fmt.lstrip(":")
        return f"{self.name} {format(self.val, spec)} ({format(self.avg, spec)})"


class ProgressMeter:
    def __init__(self, num_batches, meters, prefix=""):
        width = len(str(num_batches))
        self._fmt = "[{:" + str(width) + "d}/" + str(num_batches) + "]"
        self.meters, self.prefix = meters, prefix

    def display(self, batch):
        print("\t".join([self.prefix + self._fmt.format(batch)] + [str(m) for m in self.meters]))
