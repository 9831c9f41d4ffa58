"""Spatial smoothing of fluxes (tango/smoother.py:6-57), tg_moving_average kernel."""
from . import _ops


class Smoother(object):
    def __init__(self, windowSize):
        self.windowSize = add_one_if_even(windowSize)

    def smooth(self, x):
        return moving_average(x, self.windowSize)


def moving_average(x, windowSize):
    """Centred moving average; the (windowSize-1)/2 points at either end use the largest symmetric
    window that fits.  windowSize must be odd; 1 returns the input unchanged."""
    if windowSize == 1:
        return x
    assert windowSize % 2 == 1, 'windowSize must be an odd integer.'
    return _ops.moving_average(x, int(windowSize))


def add_one_if_even(n):
    if n % 2 == 0:
        return n + 1
    elif n % 2 == 1:
        return n
    raise ValueError('n does not appear to be an integer.')
