"""Moving profiles and transport coefficients between the transport grid and a turbulence grid that covers part of
it (tango/interfacegrids_gene.py:20-88, :446-526).  The resampling is the cubic interpolating spline of
scipy.interpolate.InterpolatedUnivariateSpline, evaluated on the GPU (tg_spline_resample); arrays may be (N,) or
(profiles, N).  TangoOutsideExtrapCoeffs adds the polynomial extrapolation of the coefficients beyond the trusted part
of the turbulence grid (a least-squares fit over a handful of points, done on the host).  The two-sided
extrapolation and the TangoInside family of the reference are not part of this package yet."""
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
    if fIn.ndim == 1:
        fOut[right] = np.polyval(np.polyfit(xIn[zone], fIn[zone], polynomialDegree), xOut[right])
    else:
        coef = np.polyfit(xIn[zone], fIn[:, zone].T, polynomialDegree)          # (degree + 1, profiles)
        fOut[:, right] = np.stack([np.polyval(coef[:, b], xOut[right]) for b in range(fIn.shape[0])])
    if enforcePositive:
        fOut[fOut < 0] = 0
    return fOut


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
