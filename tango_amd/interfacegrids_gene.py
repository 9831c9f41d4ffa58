"""Moving profiles and transport coefficients between the transport grid and a turbulence grid that covers part of
it (tango/interfacegrids_gene.py:20-88, :446-526).  The resampling is the cubic interpolating spline of
scipy.interpolate.InterpolatedUnivariateSpline, evaluated on the GPU (tg_spline_resample); arrays may be (N,) or
(profiles, N).  TangoOutsideExtrapCoeffs adds the polynomial extrapolation of the coefficients beyond the trusted part
of the turbulence grid (a least-squares fit over a handful of points, done on the host), the BothSides variant does
so at the inner boundary as well (:181-245, :300-364).  The TangoInside family (:367-443, :528-692) is for a
turbulence grid that reaches beyond the outer end of the transport grid: the profile is continued there by a
straight line (fitted slope or imposed slope), the coefficients are cut off."""
import numpy as np

from . import _ops
from .multifield import GridsNull  # noqa: F401  (tango/interfacegrids_gene.py:20-33)


class GridInterfaceTangoOutside(object):
    """The transport grid extends beyond the turbulence grid at both ends: profiles are resampled onto the
    turbulence grid with the interpolating spline; transport coefficients are resampled onto the overlapping part
    of the transport grid and are zero outside (tango/interfacegrids_gene.py:35-87)."""

    def __init__(self, psiTango, psiGene):
        psiTango, psiGene = np.asarray(psiTango, dtype=np.float64), np.asarray(psiGene, dtype=np.float64)
        assert psiTango[-1] >= psiGene[-1]
        self.psiTango = psiTango
        self.psiGene = psiGene

    def map_profile_onto_turb_grid(self, profileTango):
        return _ops.spline_resample(self.psiTango, profileTango, self.psiGene)

    def map_transport_coeffs_onto_transport_grid(self, DGeneGrid, cGeneGrid):
        return (self.map_to_transport_grid(DGeneGrid, enforcePositive=True), self.map_to_transport_grid(cGeneGrid))

    def map_to_transport_grid(self, fGene, enforcePositive=False):
        return extend_with_zeros_both_sides(self.psiGene, fGene, self.psiTango, enforcePositive=enforcePositive)

    def get_x_transport_grid(self):
        return self.psiTango

    def get_x_turbulence_grid(self):
        return self.psiGene


class TangoOutsideExtrapCoeffs(object):
    """As GridInterfaceTangoOutside, but the coefficients right of xExtrapZoneRight (the turbulence code's buffer
    zone and beyond) are replaced by a polynomial fitted to the turbulence-grid data inside
    [xExtrapZoneLeft, xExtrapZoneRight] (tango/interfacegrids_gene.py:90-178)."""

    def __init__(self, xTango, xTurb, xExtrapZoneLeft, xExtrapZoneRight, polynomialDegree):
        xTango, xTurb = np.asarray(xTango, dtype=np.float64), np.asarray(xTurb, dtype=np.float64)
        assert xTango[-1] >= xTurb[-1]
        self.xTango, self.xTurb = xTango, xTurb
        self.xExtrapZoneLeft, self.xExtrapZoneRight = xExtrapZoneLeft, xExtrapZoneRight
        self.polynomialDegree = polynomialDegree

    def map_profile_onto_turb_grid(self, profileTango):
        return _ops.spline_resample(self.xTango, profileTango, self.xTurb)

    def map_transport_coeffs_onto_transport_grid(self, DTurbGrid, cTurbGrid):
        return (self.map_to_transport_grid(DTurbGrid, enforcePositive=True), self.map_to_transport_grid(cTurbGrid))

    def map_to_transport_grid(self, fTurb, enforcePositive=False):
        return zeropad_on_left_extrap_on_right(self.xTurb, fTurb, self.xTango, self.xExtrapZoneLeft, self.xExtrapZoneRight,
                                               self.polynomialDegree, enforcePositive=enforcePositive)

    def get_x_transport_grid(self):
        return self.xTango

    def get_x_turbulence_grid(self):
        return self.xTurb


class TangoOutsideExtrapCoeffsBothSides(object):
    """As TangoOutsideExtrapCoeffs with a second extrapolation zone at the inner boundary: the coefficients left of
    xInnerExtrapZoneLeft come from a polynomial fitted inside [xInnerExtrapZoneLeft, xInnerExtrapZoneRight]
    (tango/interfacegrids_gene.py:181-245)."""

    def __init__(self, xTango, xTurb, xInnerExtrapZoneLeft, xInnerExtrapZoneRight, xOuterExtrapZoneLeft,
                 xOuterExtrapZoneRight, polynomialDegree):
        xTango, xTurb = np.asarray(xTango, dtype=np.float64), np.asarray(xTurb, dtype=np.float64)
        assert xTango[-1] >= xTurb[-1]
        self.xTango, self.xTurb = xTango, xTurb
        self.xInnerExtrapZoneLeft, self.xInnerExtrapZoneRight = xInnerExtrapZoneLeft, xInnerExtrapZoneRight
        self.xOuterExtrapZoneLeft, self.xOuterExtrapZoneRight = xOuterExtrapZoneLeft, xOuterExtrapZoneRight
        self.polynomialDegree = polynomialDegree

    def map_profile_onto_turb_grid(self, profileTango):
        return _ops.spline_resample(self.xTango, profileTango, self.xTurb)

    def map_transport_coeffs_onto_transport_grid(self, DTurbGrid, cTurbGrid):
        return (self.map_to_transport_grid(DTurbGrid, enforcePositive=True), self.map_to_transport_grid(cTurbGrid))

    def map_to_transport_grid(self, fTurb, enforcePositive=False):
        return extrap_on_left_and_right(self.xTurb, fTurb, self.xTango, self.xInnerExtrapZoneLeft,
                                        self.xInnerExtrapZoneRight, self.xOuterExtrapZoneLeft, self.xOuterExtrapZoneRight,
                                        self.polynomialDegree, enforcePositive=enforcePositive)

    def get_x_transport_grid(self):
        return self.xTango

    def get_x_turbulence_grid(self):
        return self.xTurb


def _polyfit_eval(xIn, fIn, zone, degree, xEval):
    """Least-squares polynomial through the input points marked by `zone`, evaluated at xEval, for (N,) or
    (profiles, N) data (np.polyfit / np.polyval as in tango/interfacegrids_gene.py:288-292)."""
    if fIn.ndim == 1:
        return np.polyval(np.polyfit(xIn[zone], fIn[zone], degree), xEval)
    coef = np.polyfit(xIn[zone], fIn[:, zone].T, degree)          # (degree + 1, profiles)
    return np.stack([np.polyval(coef[:, b], xEval) for b in range(fIn.shape[0])])


def zeropad_on_left_extrap_on_right(xIn, fIn, xOut, xExtrapZoneLeft, xExtrapZoneRight, polynomialDegree, enforcePositive=False):
    """Zeros left of xIn[0], the interpolating spline on xIn[0] <= x <= xExtrapZoneRight, and right of that the
    polynomial of the given degree fitted (least squares) to the input points inside the extrapolation zone
    (tango/interfacegrids_gene.py:248-298)."""
    xIn, xOut = np.asarray(xIn, dtype=np.float64), np.asarray(xOut, dtype=np.float64)
    assert xOut[0] <= xIn[0] and xOut[-1] >= xIn[-1]
    inside = np.flatnonzero((xOut >= xIn[0]) & (xOut <= xExtrapZoneRight))
    lo, hi = (int(inside[0]), int(inside[-1]) + 1) if inside.size else (0, 0)
    fIn = np.asarray(fIn, dtype=np.float64)
    fOut = np.array(_ops.spline_resample(xIn, fIn, xOut, lo, hi))
    zone = (xIn >= xExtrapZoneLeft) & (xIn <= xExtrapZoneRight)
    right = xOut > xExtrapZoneRight
    fOut[..., right] = _polyfit_eval(xIn, fIn, zone, polynomialDegree, xOut[right])
    if enforcePositive:
        fOut[fOut < 0] = 0
    return fOut


def extrap_on_left_and_right(xIn, fIn, xOut, xInnerExtrapZoneLeft, xInnerExtrapZoneRight, xOuterExtrapZoneLeft,
                             xOuterExtrapZoneRight, polynomialDegree, enforcePositive=False):
    """The interpolating spline on xIn[0] <= x <= xOuterExtrapZoneRight; left of xInnerExtrapZoneLeft and right of
    xOuterExtrapZoneRight the polynomials fitted inside the inner and the outer extrapolation zone; an inner fit that
    reaches into the spline's range replaces it there, as in the reference (tango/interfacegrids_gene.py:300-364)."""
    xIn, xOut = np.asarray(xIn, dtype=np.float64), np.asarray(xOut, dtype=np.float64)
    assert xOut[0] <= xIn[0] and xOut[-1] >= xIn[-1]
    inside = np.flatnonzero((xOut >= xIn[0]) & (xOut <= xOuterExtrapZoneRight))
    lo, hi = (int(inside[0]), int(inside[-1]) + 1) if inside.size else (0, 0)
    fIn = np.asarray(fIn, dtype=np.float64)
    fOut = np.array(_ops.spline_resample(xIn, fIn, xOut, lo, hi))
    left = xOut < xInnerExtrapZoneLeft
    fOut[..., left] = _polyfit_eval(xIn, fIn, (xIn >= xInnerExtrapZoneLeft) & (xIn <= xInnerExtrapZoneRight),
                                    polynomialDegree, xOut[left])
    right = xOut > xOuterExtrapZoneRight
    fOut[..., right] = _polyfit_eval(xIn, fIn, (xIn >= xOuterExtrapZoneLeft) & (xIn <= xOuterExtrapZoneRight),
                                     polynomialDegree, xOut[right])
    if enforcePositive:
        fOut[fOut < 0] = 0
    return fOut


class GridInterfaceTangoInside(object):
    """The turbulence grid reaches beyond the outer end of the transport grid, the transport grid beyond the inner
    end of the turbulence grid: the profile is continued outward by the constrained regression line, the
    coefficients are resampled where the grids overlap and zero further in (tango/interfacegrids_gene.py:367-417)."""

    def __init__(self, psiTango, psiGene):
        psiTango, psiGene = np.asarray(psiTango, dtype=np.float64), np.asarray(psiGene, dtype=np.float64)
        assert psiGene[-1] >= psiTango[-1]
        self.psiTango, self.psiGene = psiTango, psiGene

    def map_profile_onto_turb_grid(self, profileTango):
        return truncate_on_left_extrapolate_on_right(self.psiTango, profileTango, self.psiGene, enforcePositive=True)

    def map_transport_coeffs_onto_transport_grid(self, DGeneGrid, cGeneGrid):
        return (self.map_to_transport_grid(DGeneGrid, enforcePositive=True), self.map_to_transport_grid(cGeneGrid))

    def map_to_transport_grid(self, fGene, enforcePositive=False):
        return extend_with_zeros_left_side(self.psiGene, fGene, self.psiTango, enforcePositive=enforcePositive)

    def get_x_transport_grid(self):
        return self.psiTango

    def get_x_turbulence_grid(self):
        return self.psiGene


class GridInterfaceTangoInsideFixedOutside(object):
    """As GridInterfaceTangoInside with an imposed (negative) slope of the profile beyond the outer end of the
    transport grid instead of a fitted one (tango/interfacegrids_gene.py:419-443)."""

    def __init__(self, psiTango, psiGene, outwardSlopeGene):
        self.psiTango, self.psiGene = np.asarray(psiTango, dtype=np.float64), np.asarray(psiGene, dtype=np.float64)
        self.outwardSlopeGene = outwardSlopeGene
        assert outwardSlopeGene < 0, "You probably meant to impose a negative, not positive, slope on the outward side."

    def map_profile_onto_turb_grid(self, profileTango):
        return truncate_on_left_fixed_slope_on_right(self.psiTango, profileTango, self.psiGene, self.outwardSlopeGene,
                                                     enforcePositive=True)

    def map_transport_coeffs_onto_transport_grid(self, DGeneGrid, cGeneGrid):
        return (self.map_to_transport_grid(DGeneGrid, enforcePositive=True), self.map_to_transport_grid(cGeneGrid))

    def map_to_transport_grid(self, fGene, enforcePositive=False):
        return extend_with_zeros_left_side(self.psiGene, fGene, self.psiTango, enforcePositive=enforcePositive)

    def get_x_transport_grid(self):
        return self.psiTango

    def get_x_turbulence_grid(self):
        return self.psiGene


def extend_with_zeros_both_sides(xSmall, fSmall, xLarge, enforcePositive=False):
    """fSmall on xSmall resampled onto the points of xLarge strictly inside (xSmall[0], xSmall[-1]), zero at the
    others; negative values replaced by zero on request (tango/interfacegrids_gene.py:446-484)."""
    xSmall, xLarge = np.asarray(xSmall, dtype=np.float64), np.asarray(xLarge, dtype=np.float64)
    assert xLarge[0] <= xSmall[0] and xLarge[-1] >= xSmall[-1]
    lo = int(np.where(xLarge > xSmall[0])[0][0])
    hi = int(np.where(xLarge < xSmall[-1])[0][-1]) + 1
    return _ops.spline_resample(xSmall, fSmall, xLarge, lo, hi, nonneg=enforcePositive)


def extend_with_zeros_left_side(xIn, fIn, xOut, enforcePositive=False):
    """The same with an output grid that starts to the left of the input grid and ends inside it
    (tango/interfacegrids_gene.py:486-526)."""
    xIn, xOut = np.asarray(xIn, dtype=np.float64), np.asarray(xOut, dtype=np.float64)
    assert xOut[0] <= xIn[0] and xOut[-1] <= xIn[-1]
    lo = int(np.where(xOut > xIn[0])[0][0])
    return _ops.spline_resample(xIn, fIn, xOut, lo, xOut.size, nonneg=enforcePositive)


def least_squares_slope(x, y, x0, y0):
    """Slope of the least-squares line through (x0, y0): (x-x0).(y-y0) / (x-x0).(x-x0); y and y0 may carry a leading
    profile axis (tango/interfacegrids_gene.py:528-543)."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    d = x - x0
    if y.ndim == 1:
        return np.dot(d, y - y0) / np.dot(d, d)
    return (y - np.asarray(y0)[:, None]) @ d / np.dot(d, d)


def extrap1d_constrained_linear_regression(x, y, xEval, side='right', numPts=10):
    """The line through the end point of the data on the given side with the least-squares slope of the numPts
    points next to it, evaluated at xEval (tango/interfacegrids_gene.py:545-576)."""
    assert side == 'left' or side == 'right'
    x, y, xEval = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), np.asarray(xEval, dtype=np.float64)
    if side == 'left':
        xSide, ySide, x0, y0 = x[:numPts], y[..., :numPts], x[0], y[..., 0]
    else:
        xSide, ySide, x0, y0 = x[-numPts:], y[..., -numPts:], x[-1], y[..., -1]
    a = least_squares_slope(xSide, ySide, x0, y0)
    if y.ndim == 1:
        return a * (xEval - x0) + y0
    return a[:, None] * (xEval - x0)[None, :] + y0[:, None]


def _spline_strictly_inside(xSmall, fSmall, xLarge, fLarge, closed=False):
    """Overwrite the entries of fLarge whose point lies inside (xSmall[0], xSmall[-1]) (closed: end points included)
    with the interpolating spline."""
    if closed:
        ind = np.flatnonzero((xLarge >= xSmall[0]) & (xLarge <= xSmall[-1]))
    else:
        ind = np.flatnonzero((xLarge > xSmall[0]) & (xLarge < xSmall[-1]))
    if ind.size:
        lo, hi = int(ind[0]), int(ind[-1]) + 1
        fLarge[..., lo:hi] = np.asarray(_ops.spline_resample(xSmall, fSmall, xLarge, lo, hi))[..., lo:hi]
    return fLarge


def make_extrapolator(xSmall, fSmall, side, numPts):
    """Callable: cubic interpolation strictly inside the data, constrained linear regression of the numPts points on
    `side` everywhere else (tango/interfacegrids_gene.py:579-604)."""
    xSmall, fSmall = np.asarray(xSmall, dtype=np.float64), np.asarray(fSmall, dtype=np.float64)

    def extrapolator(xLarge):
        xLarge = np.asarray(xLarge, dtype=np.float64)
        fLarge = extrap1d_constrained_linear_regression(xSmall, fSmall, xLarge, side, numPts=numPts)
        return _spline_strictly_inside(xSmall, fSmall, xLarge, fLarge)
    return extrapolator


def truncate_on_left_extrapolate_on_right(xIn, fIn, xOut, numPts=10, enforcePositive=False):
    """fIn moved to a grid that starts inside xIn and ends to the right of it: spline where the grids overlap, the
    regression line beyond (tango/interfacegrids_gene.py:606-634)."""
    xIn, xOut = np.asarray(xIn, dtype=np.float64), np.asarray(xOut, dtype=np.float64)
    assert xOut[0] >= xIn[0] and xOut[-1] >= xIn[-1]
    fOut = make_extrapolator(xIn, fIn, side='right', numPts=numPts)(xOut)
    if enforcePositive:
        fOut[fOut < 0] = 0
    return fOut


def make_extrapolator_fixed_slope(xSmall, fSmall, outwardSlope):
    """Callable: cubic interpolation on the closed range of the data, straight lines of the imposed slope through the
    end points outside (tango/interfacegrids_gene.py:636-662)."""
    xSmall, fSmall = np.asarray(xSmall, dtype=np.float64), np.asarray(fSmall, dtype=np.float64)

    def extrapolator(xLarge):
        xLarge = np.asarray(xLarge, dtype=np.float64)
        fLarge = np.zeros(fSmall.shape[:-1] + xLarge.shape)
        left, right = xLarge < xSmall[0], xLarge > xSmall[-1]
        fLarge[..., left] = outwardSlope * (xLarge[left] - xSmall[0]) + fSmall[..., :1]
        fLarge[..., right] = outwardSlope * (xLarge[right] - xSmall[-1]) + fSmall[..., -1:]
        return _spline_strictly_inside(xSmall, fSmall, xLarge, fLarge, closed=True)
    return extrapolator


def truncate_on_left_fixed_slope_on_right(xIn, fIn, xOut, outwardSlope, enforcePositive=False):
    """As truncate_on_left_extrapolate_on_right with the imposed slope beyond the right end of xIn
    (tango/interfacegrids_gene.py:664-692)."""
    xIn, xOut = np.asarray(xIn, dtype=np.float64), np.asarray(xOut, dtype=np.float64)
    assert xOut[0] >= xIn[0] and xOut[-1] >= xIn[-1]
    fOut = make_extrapolator_fixed_slope(xIn, fIn, outwardSlope)(xOut)
    if enforcePositive:
        fOut[fOut < 0] = 0
    return fOut
