"""Centred finite differences along the radial grid (tango/derivatives.py:9-41), tg_ddx kernel."""
from . import _ops


def dx_centered_difference_edge_first_order(u, dx):
    """Two-point one-sided differences at both ends."""
    return _ops.ddx(u, dx, 1)


def dx_centered_difference(u, dx):
    """Three-point second-order differences at both ends."""
    return _ops.ddx(u, dx, 2)
