"""Progress-text sink (reference: _out.py:1-9): stdout when asked, otherwise swallow."""
import sys


class _Quiet:
    def write(self, *a, **k):
        return 0

    def flush(self):
        pass


def select_output(allow=False):
    return sys.stdout if allow else _Quiet()
