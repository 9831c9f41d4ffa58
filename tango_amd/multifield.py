"""Data model of the transport solve: fields, H-coefficient containers and the callable that gathers the
H coefficients of every field for one iteration.  Same classes, attributes and error behaviour as
tango/multifield.py:10-221; profiles and coefficients may be (N,) arrays (one profile, as in the reference)
or (profiles, N) arrays (an ensemble), and a rightBC may then be one value per profile."""
import numpy as np

_H_NAMES = ('H1', 'H2', 'H3', 'H4', 'H6', 'H7', 'H8')
_MAPPER_METHODS = ('map_profile_onto_turb_grid', 'map_transport_coeffs_onto_transport_grid')


class GridsNull(object):
    """Identity mapping between the transport grid and the turbulence grid (tango/multifield.py:211-221)."""

    def map_profile_onto_turb_grid(self, profile):
        return profile

    def map_transport_coeffs_onto_transport_grid(self, D, c):
        return (D, c)


class Field(object):
    """One transported quantity (tango/multifield.py:10-29)."""

    def __init__(self, label=None, rightBC=None, profile_mminus1=None, coupledTo=None, compute_all_H=None,
                 gridMapper=None, lodestroMethod=None):
        if label is None:
            raise ValueError('label is a required input when creating a Field.')
        if gridMapper is None:
            gridMapper = GridsNull()
        else:
            for m in _MAPPER_METHODS:
                assert callable(getattr(gridMapper, m, None)), f'gridMapper needs a callable {m}'
        self.label = label
        self.rightBC = rightBC
        self.profile_mminus1 = profile_mminus1
        self.coupledTo = coupledTo
        self.compute_all_H = compute_all_H
        self.gridMapper = gridMapper
        self.lodestroMethod = lodestroMethod


def get_field_by_label(fields, label):
    """First field carrying `label`; ValueError when there is none (tango/multifield.py:31-46)."""
    for f in fields:
        if f.label == label:
            return f
    raise ValueError('field corresponding to the given label is not found.')


def check_fields_label(fields):
    labels = [f.label for f in fields]
    return len(set(labels)) == len(labels)


def _is_bc(v, profile):
    """A right boundary value: a scalar, or for an ensemble one value per profile."""
    if np.isscalar(v):
        return True
    v = np.asarray(v)
    return profile is not None and np.ndim(profile) == 2 and v.shape == (np.shape(profile)[0],)


def check_fields_rightBC(fields):
    return all(_is_bc(f.rightBC, f.profile_mminus1) for f in fields)


def check_fields_profile_mminus1(fields):
    shape = np.shape(fields[0].profile_mminus1)
    return all(f.profile_mminus1 is not None and np.shape(f.profile_mminus1) == shape for f in fields) and len(shape) > 0


def check_fields_coupled_to(fields):
    by_label = {}
    for f in fields:
        by_label.setdefault(f.label, f)
    for f in fields:
        if f.coupledTo is None:
            continue
        partner = by_label.get(f.coupledTo)
        if f.coupledTo == f.label or partner is None or partner.coupledTo != f.label:
            return False
    return True


def check_fields_compute_all_H(fields):
    return all(callable(getattr(f, 'compute_all_H')) for f in fields)


def check_fields_initialize(fields):
    """Validate a list of fields once, raising ValueError on the first failed check (tango/multifield.py:92-106)."""
    checks = ((check_fields_label, 'labels of fields are not unique.'),
              (check_fields_rightBC, 'rightBC must be a scalar for each field.'),
              (check_fields_profile_mminus1, 'profile_mminus1 must have the same length for all fields.'),
              (check_fields_coupled_to, 'all fields must either be coupledTo None or a partner.'),
              (check_fields_compute_all_H, 'all fields must have a callable compute_all_H.'))
    for fn, msg in checks:
        if not fn(fields):
            raise ValueError(msg)


class HCoefficients(object):
    """Container of H1..H8 (no H5); None stands for an absent (zero) coefficient (tango/multifield.py:108-152)."""

    def __init__(self, H1=None, H2=None, H3=None, H4=None, H6=None, H7=None, H8=None):
        self.H1, self.H2, self.H3, self.H4, self.H6, self.H7, self.H8 = H1, H2, H3, H4, H6, H7, H8

    @staticmethod
    def _add_H(Ha, Hb):
        if Ha is None:
            return Hb
        if Hb is None:
            return Ha
        return Ha + Hb

    def __add__(self, other):
        return HCoefficients(**{k: self._add_H(getattr(self, k), getattr(other, k)) for k in _H_NAMES})

    def __repr__(self):
        body = ''.join('{}={},\n'.format(k, getattr(self, k)) for k in _H_NAMES if getattr(self, k) is not None)
        return 'HCoefficients(\n' + body + ')'


class ComputeAllHAllFields(object):
    """Callable (t, x, profiles) -> (HCoeffsAllFields, HCoeffsTurbAllFields, extradataAllFields): runs the
    turbulence handler unless frozen turbulent coefficients are passed in, then every field's own
    compute_all_H (tango/multifield.py:157-206)."""

    def __init__(self, fields, turbhandler=None):
        self.fields = fields
        self.turbhandler = turbhandler

    def __call__(self, t, x, profiles, computeTurbulence=True, HCoeffsTurbAllFields=None):
        extradataAllFields = None
        if computeTurbulence:
            if self.turbhandler is not None:
                (HCoeffsTurbAllFields, extradataAllFields) = \
                    self.turbhandler.turbflux_to_Hcoeffs_multifield(self.fields, profiles)
        else:
            assert HCoeffsTurbAllFields is not None, \
                'must provide HCoeffsTurbAllFields when using computeTurbulence==False'
        HCoeffsAllFields = {}
        for field in self.fields:
            if self.turbhandler is not None:
                HCoeffsAllFields[field.label] = field.compute_all_H(t, x, profiles, HCoeffsTurbAllFields[field.label])
            else:
                HCoeffsAllFields[field.label] = field.compute_all_H(t, x, profiles)
        return (HCoeffsAllFields, HCoeffsTurbAllFields, extradataAllFields)
