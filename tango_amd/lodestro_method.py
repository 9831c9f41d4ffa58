"""The LoDestro method: relax profiles and transport coefficients with exponentially weighted moving
averages (EWMA) over the Picard iterates and represent a turbulent flux as a diffusive plus a convective
part, flux = -D dp/dx + c p, so that the transport equation can be advanced implicitly.

Same classes and methods as tango/lodestro_method.py (TurbulenceHandler :266-408, lm :411-529,
FluxSplit :535-679, EWMA :682-728, GridsNull :730-743, and the older flux-relaxing variants
TurbulenceHandler_old / lm_old :30-260).  The arithmetic runs in the tg_ewma and tg_flux_split kernels;
arrays may be (N,) or (profiles, N)."""
import numpy as np

from . import _ops, multifield


class EWMA(object):
    """y_bar_l = A y_l + (1 - A) y_bar_{l-1}; the first iterate seeds y_bar_{l-1} = y_l."""

    def __init__(self, EWMAParam):
        self.EWMAParam = EWMAParam
        self._value = None
        self._fetch = None

    # The memory y_bar_{l-1}.  The fused solver paths hand over a callable that reads it from the device; it is
    # called the first time someone looks (an ensemble's memories are hundreds of megabytes that most runs never read).
    @property
    def _yEWMA_lminus1(self):
        if self._fetch is not None:
            self._value, self._fetch = self._fetch(), None
        return self._value

    @_yEWMA_lminus1.setter
    def _yEWMA_lminus1(self, y):
        if callable(y):
            self._value, self._fetch = None, y
        else:
            self._value, self._fetch = y, None

    def next_ewma_iterate(self, y_l):
        if np.ndim(y_l) == 0:   # scalars: plain host arithmetic
            prev = y_l if self._yEWMA_lminus1 is None else self._yEWMA_lminus1
            self._yEWMA_lminus1 = self._compute_next_ewma(y_l, prev, self.EWMAParam)
        else:
            self._yEWMA_lminus1 = _ops.ewma(y_l, self._yEWMA_lminus1, self.EWMAParam)
        return self._yEWMA_lminus1

    @staticmethod
    def _compute_next_ewma(y_l, yEWMA_lminus1, EWMAParam):
        if np.ndim(y_l) == 0 and np.ndim(yEWMA_lminus1) == 0:
            return EWMAParam * y_l + (1 - EWMAParam) * yEWMA_lminus1
        return _ops.ewma(y_l, yEWMA_lminus1, EWMAParam)

    def reset_ewma_iterate(self):
        self._yEWMA_lminus1 = None

    def set_ewma_iterate(self, yEWMA):
        self._yEWMA_lminus1 = yEWMA


class FluxSplit(object):
    """Split a flux into D = theta*DHat and c = (1-theta)*cHat with DHat = -flux/(<g^xx> dp/dx),
    cHat = flux/(<|grad x|> p) and 0 <= theta <= 1 (tango/lodestro_method.py:563-679).

    The default theta rule runs on the GPU.  A callable thetaParams['custom_ftheta'](DHat, dpdx, thetaParams)
    is honoured as in the reference; it is evaluated on the host between two kernels."""

    def __init__(self, thetaParams, gxxAvg=1, gradxAvg=1):
        self.thetaParams = thetaParams
        self.gxxAvg = gxxAvg
        self.gradxAvg = gradxAvg
        custom = thetaParams.get('custom_ftheta') if hasattr(thetaParams, 'get') else None
        self.ftheta = custom if callable(custom) else self._default_ftheta

    def flux_to_transport_coeffs(self, flux, p, dx):
        tp = self.thetaParams
        if self.ftheta is not self._default_ftheta:
            return self._with_custom_theta(flux, p, dx)
        D, c, DHat, cHat, theta = _ops.flux_split(flux, p, dx, tp['Dmin'], tp['Dmax'], tp['dpdxThreshold'],
                                                  self.gxxAvg, self.gradxAvg)
        if not hasattr(theta, 'to_host'):
            assert np.count_nonzero((theta >= 0) & (theta <= 1)) == np.size(theta), 'some theta is not between 0 and 1'
        return (D, c, {'DHat': DHat, 'cHat': cHat, 'theta': theta})

    def _with_custom_theta(self, flux, p, dx):
        # DHat, cHat and dp/dx from the kernels (theta parameters irrelevant), theta from the user's callable
        _, _, DHat, cHat, _ = _ops.flux_split(flux, p, dx, 0.0, 1.0, 0.0, self.gxxAvg, self.gradxAvg)
        dpdx = _ops.ddx(p, dx, 1)
        theta = self.ftheta(DHat, dpdx, self.thetaParams)
        return (theta * DHat, (1 - theta) * cHat, {'DHat': DHat, 'cHat': cHat, 'theta': theta})

    @staticmethod
    def _default_ftheta(DHat, dpdx, thetaParams):
        """Host statement of the default rule for callers that use it directly: theta = 0 below Dmin;
        for |dp/dx| below the threshold a linear ramp from 1 at Dmin to 1/2 at Dmax and 1/2 above Dmax;
        1 otherwise."""
        Dmin, Dmax, thr = thetaParams['Dmin'], thetaParams['Dmax'], thetaParams['dpdxThreshold']
        DHat = np.asarray(DHat, dtype=float)
        small = np.abs(dpdx) < thr
        theta = np.where(DHat < Dmin, 0.0, 1.0)
        ramp = small & (DHat >= Dmin) & (DHat <= Dmax)
        theta = np.where(ramp, (1 + (Dmax - DHat) / (Dmax - Dmin)) / 2, theta)
        theta = np.where(small & (DHat > Dmax), 0.5, theta)
        assert np.count_nonzero((theta >= 0) & (theta <= 1)) == np.size(theta), 'some theta is not between 0 and 1'
        return theta


class lm(object):
    """Per-field LoDestro state: EWMA memories of D, c and of the profile, and the flux splitter."""

    def __init__(self, EWMAParamTurbFlux, EWMAParamProfile, thetaParams, gxxAvgTurb=1, gradxAvgTurb=1):
        self._EWMATurbD = EWMA(EWMAParamTurbFlux)
        self._EWMATurbc = EWMA(EWMAParamTurbFlux)
        self._EWMAProfile = EWMA(EWMAParamProfile)
        self._fluxSplitter = FluxSplit(thetaParams, gxxAvgTurb, gradxAvgTurb)

    def ewma_turb_D(self, D_l):
        return self._EWMATurbD.next_ewma_iterate(D_l)

    def ewma_turb_c(self, c_l):
        return self._EWMATurbc.next_ewma_iterate(c_l)

    def ewma_profile(self, profile_l):
        return self._EWMAProfile.next_ewma_iterate(profile_l)

    def flux_to_transport_coeffs(self, flux, p, dx):
        return self._fluxSplitter.flux_to_transport_coeffs(flux, p, dx)

    def get_ewma_params(self):
        return (self._EWMATurbD.EWMAParam, self._EWMAProfile.EWMAParam)

    def set_ewma_params(self, EWMAParamTurbFlux, EWMAParamProfile):
        self._EWMATurbD.EWMAParam = EWMAParamTurbFlux
        self._EWMATurbc.EWMAParam = EWMAParamTurbFlux
        self._EWMAProfile.EWMAParam = EWMAParamProfile

    def set_ewma_iterates(self, profileEWMA, turbDEWMA, turbcEWMA):
        self.set_ewma_profile(profileEWMA)
        self.set_ewma_turb_D(turbDEWMA)
        self.set_ewma_turb_c(turbcEWMA)

    def set_ewma_profile(self, profileEWMA):
        self._EWMAProfile.set_ewma_iterate(profileEWMA)

    def set_ewma_turb_D(self, turbDEWMA):
        self._EWMATurbD.set_ewma_iterate(turbDEWMA)

    def set_ewma_turb_c(self, turbcEWMA):
        self._EWMATurbc.set_ewma_iterate(turbcEWMA)


class TurbulenceHandler(object):
    """Turbulent fluxes -> H2, H3 contributions of every field (tango/lodestro_method.py:266-408).

    fluxModel is any object with get_flux(dict of profiles on the turbulence grid) -> dict of fluxes."""

    def __init__(self, dxTurbGrid, xTango, fluxModel, VprimeTango=None, fluxSmoother=None,
                 gxxAvgTango=1, gradxAvgTango=1):
        assert callable(getattr(fluxModel, 'get_flux', None))
        self.dxTurbGrid = dxTurbGrid
        self.xTango = xTango
        self.fluxModel = fluxModel
        self.isNonCartesian = VprimeTango is not None
        if self.isNonCartesian:
            self.VprimeTango = VprimeTango
        self.doSmoothing = fluxSmoother is not None
        if self.doSmoothing:
            assert callable(getattr(fluxSmoother, 'smooth', None))
            self.fluxSmoother = fluxSmoother
        self.gxxAvgTango = gxxAvgTango
        self.gradxAvgTango = gradxAvgTango

    def turbflux_to_Hcoeffs_multifield(self, fields, profiles):
        onTurb, relaxed = {}, {}
        for field in fields:
            onTurb[field.label] = field.gridMapper.map_profile_onto_turb_grid(profiles[field.label])
            relaxed[field.label] = field.lodestroMethod.ewma_profile(onTurb[field.label])
        fluxes = self.fluxModel.get_flux(relaxed)
        HCoeffsTurbAllFields, extradataAllFields = {}, {}
        for field in fields:
            label, method = field.label, field.lodestroMethod
            smoothed = self.fluxSmoother.smooth(fluxes[label]) if self.doSmoothing else fluxes[label]
            (DTurb, cTurb, split) = method.flux_to_transport_coeffs(smoothed, relaxed[label], self.dxTurbGrid)
            DEWMA = method.ewma_turb_D(DTurb)
            cEWMA = method.ewma_turb_c(cTurb)
            (D, c) = field.gridMapper.map_transport_coeffs_onto_transport_grid(DEWMA, cEWMA)
            (H2contrib, H3contrib) = self.Dc_to_Hcontrib(D, c)
            HCoeffsTurbAllFields[label] = multifield.HCoefficients(H2=H2contrib, H3=H3contrib)
            extradataAllFields[label] = {
                'D': D, 'c': c, 'profileTurbGrid': onTurb[label], 'profileEWMATurbGrid': relaxed[label],
                'fluxTurbGrid': fluxes[label], 'smoothedFluxTurbGrid': smoothed, 'DTurbGrid': DTurb, 'cTurbGrid': cTurb,
                'DEWMATurbGrid': DEWMA, 'cEWMATurbGrid': cEWMA, 'DHatTurbGrid': split['DHat'],
                'cHatTurbGrid': split['cHat'], 'thetaTurbGrid': split['theta']}
        return (HCoeffsTurbAllFields, extradataAllFields)

    def Dc_to_Hcontrib(self, D, c):
        """H2 = V' D <g^xx>, H3 = -V' c <|grad x|>; H2 = D, H3 = -c in Cartesian geometry."""
        if self.isNonCartesian:
            return (self.VprimeTango * D * self.gxxAvgTango, -self.VprimeTango * c * self.gradxAvgTango)
        return (D, -c)


class lm_old(object):
    """Variant that relaxes the turbulent flux itself instead of D and c (tango/lodestro_method.py:167-260)."""

    def __init__(self, EWMAParamTurbFlux, EWMAParamProfile, thetaParams):
        self._EWMATurbFlux = EWMA(EWMAParamTurbFlux)
        self._EWMAProfile = EWMA(EWMAParamProfile)
        self._fluxSplitter = FluxSplit(thetaParams)

    def ewma_turb_flux(self, turbFlux_l):
        return self._EWMATurbFlux.next_ewma_iterate(turbFlux_l)

    def ewma_profile(self, profile_l):
        return self._EWMAProfile.next_ewma_iterate(profile_l)

    def flux_to_transport_coeffs(self, flux, p, dx):
        return self._fluxSplitter.flux_to_transport_coeffs(flux, p, dx)

    def get_ewma_params(self):
        return (self._EWMATurbFlux.EWMAParam, self._EWMAProfile.EWMAParam)

    def set_ewma_params(self, EWMAParamTurbFlux, EWMAParamProfile):
        self._EWMATurbFlux.EWMAParam = EWMAParamTurbFlux
        self._EWMAProfile.EWMAParam = EWMAParamProfile

    def set_ewma_iterates(self, profileEWMA, turbFluxEWMA):
        self.set_ewma_profile(profileEWMA)
        self.set_ewma_turb_flux(turbFluxEWMA)

    def set_ewma_profile(self, profileEWMA):
        self._EWMAProfile.set_ewma_iterate(profileEWMA)

    def set_ewma_turb_flux(self, turbFluxEWMA):
        self._EWMATurbFlux.set_ewma_iterate(turbFluxEWMA)


class TurbulenceHandler_old(object):
    """Flux-relaxing variant of TurbulenceHandler for lm_old (tango/lodestro_method.py:30-164)."""

    def __init__(self, dxTurbGrid, xTango, fluxModel, VprimeTango=None, fluxSmoother=None):
        assert callable(getattr(fluxModel, 'get_flux', None))
        self.dxTurbGrid, self.xTango, self.fluxModel = dxTurbGrid, xTango, fluxModel
        self.isNonCartesian = VprimeTango is not None
        if self.isNonCartesian:
            self.VprimeTango = VprimeTango
        self.doSmoothing = fluxSmoother is not None
        if self.doSmoothing:
            assert callable(getattr(fluxSmoother, 'smooth', None))
            self.fluxSmoother = fluxSmoother

    def turbflux_to_Hcoeffs_multifield(self, fields, profiles):
        onTurb, relaxed = {}, {}
        for field in fields:
            onTurb[field.label] = field.gridMapper.map_profile_onto_turb_grid(profiles[field.label])
            relaxed[field.label] = field.lodestroMethod.ewma_profile(onTurb[field.label])
        fluxes = self.fluxModel.get_flux(relaxed)
        HCoeffsTurbAllFields, extradataAllFields = {}, {}
        for field in fields:
            label, method = field.label, field.lodestroMethod
            smoothed = self.fluxSmoother.smooth(fluxes[label]) if self.doSmoothing else fluxes[label]
            fluxEWMA = method.ewma_turb_flux(smoothed)
            (DTurb, cTurb, split) = method.flux_to_transport_coeffs(fluxEWMA, relaxed[label], self.dxTurbGrid)
            (D, c) = field.gridMapper.map_transport_coeffs_onto_transport_grid(DTurb, cTurb)
            (H2contrib, H3contrib) = self.Dc_to_Hcontrib(D, c)
            HCoeffsTurbAllFields[label] = multifield.HCoefficients(H2=H2contrib, H3=H3contrib)
            extradataAllFields[label] = {
                'D': D, 'c': c, 'profileTurbGrid': onTurb[label], 'profileEWMATurbGrid': relaxed[label],
                'fluxTurbGrid': fluxes[label], 'smoothedFluxTurbGrid': smoothed, 'fluxEWMATurbGrid': fluxEWMA,
                'DTurbGrid': DTurb, 'cTurbGrid': cTurb, 'DHatTurbGrid': split['DHat'], 'cHatTurbGrid': split['cHat'],
                'thetaTurbGrid': split['theta']}
        return (HCoeffsTurbAllFields, extradataAllFields)

    def Dc_to_Hcontrib(self, D, c):
        if self.isNonCartesian:
            return (self.VprimeTango * D, -self.VprimeTango * c)
        return (D, -c)


class GridsNull(object):
    """Single-grid mapper that also reports the grid (tango/lodestro_method.py:730-743)."""

    def __init__(self, x):
        self.x = x

    def map_profile_onto_turb_grid(self, profile):
        return profile

    def map_transport_coeffs_onto_transport_grid(self, D, c):
        return (D, c)

    def get_x_transport_grid(self):
        return self.x

    def get_x_turbulence_grid(self):
        return self.x
