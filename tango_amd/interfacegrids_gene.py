"""Moving profiles and transport coefficients between the transport grid and a turbulence grid that covers part of
it (tango/interfacegrids_gene.py:20-88, :446-526).  The resampling is the cubic interpolating spline of
scipy.interpolate.InterpolatedUnivariateSpline, evaluated on the GPU (tg_spline_resample); arrays may be (N,) or
(profiles, N).  The classes with polynomial extrapolation of the coefficients (TangoOutsideExtrapCoeffs and the
TangoInside family of the reference) are not part of this package yet."""
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
