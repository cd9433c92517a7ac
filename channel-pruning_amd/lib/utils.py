"""The helpers of the reference's lib/utils.py that the pruning path uses: Timer
(utils.py:99-123, the only profiler the reference has), printstage (13-23), underline and
CHECK_EQ (75-82)."""
import time

import numpy as np

cnt = 0


def printstage(*sentence):
    global cnt
    print('~' * 58)
    print("stage" + str(cnt) + " " + ''.join(str(i) for i in sentence))
    print('~' * 58)
    cnt += 1


def underline(*parts):
    return '_'.join(parts)


def CHECK_EQ(fake, real, tol=1e-4):
    """|fake - real| <= tol elementwise (utils.py:75-82)."""
    diff = np.max(np.abs(np.asarray(fake) - np.asarray(real)))
    assert diff <= tol, "CHECK_EQ failed: max abs diff %g" % diff
    return True


class Timer(object):
    """tic/toc wall-clock timer with the reference's interface."""

    def __init__(self):
        self.total_time = 0.
        self.calls = 0
        self.start_time = 0.
        self.diff = 0.
        self.average_time = 0.

    def tic(self):
        self.start_time = time.time()

    def toc(self, show=None, average=False):
        self.diff = time.time() - self.start_time
        self.total_time += self.diff
        self.calls += 1
        self.average_time = self.total_time / self.calls
        if show is not None:
            print(show, self.diff)
        return self.average_time if average else self.diff
