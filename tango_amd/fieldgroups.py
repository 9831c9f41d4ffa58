"""Grouping of fields into independent linear solves: a field on its own is a scalar tridiagonal system,
two fields coupled to each other form one 2x2 block-tridiagonal system.  Same classes and functions as
tango/fieldgroups.py:30-346; assembly, residual and solve are HIP kernels."""
from collections import namedtuple

from . import HToMatrixFD, JK_to_matrix_fd, _ops, multifield

UncoupledMatrixEqn = namedtuple('UncoupledMatrixEquation', ('A', 'B', 'C', 'f'))
PairCoupledMatrixEqn = namedtuple('PairCoupledMatrixEqn', ('UEqnCoeffs', 'WEqnCoeffs'))


class FieldGroup(object):
    """Common attributes of a group: the matrix equation of the current iteration, the H coefficients it was
    built from, right boundary value(s), previous-timestep profile(s) and the solution."""

    def __init__(self, label=None):
        self.matrixEqn = None
        self.HCoeffs = None
        self.profileSolution = None
        self.rightBC = None
        self.psi_mminus1 = None
        self.fieldsolution = None

    def Hcoeffs_to_matrix_eqn(self, dt, dx, rightBC, psi_mminus1, HCoeffs):
        raise NotImplementedError

    def solve_matrix_eqn(self, matrixEqn):
        raise NotImplementedError

    def calculate_residual(self, matrixEqn, profiles):
        """r = M[u] - f with the matrix built from the current iterate, divided by max|f|."""
        raise NotImplementedError


class UncoupledFieldGroup(FieldGroup):
    """tango/fieldgroups.py:71-144."""

    def __init__(self, label):
        FieldGroup.__init__(self)
        self.label = label

    def Hcoeffs_to_matrix_eqn(self, dt, dx, rightBC, psi_mminus1, HCoeffs):
        return UncoupledMatrixEqn(*HToMatrixFD.H_to_matrix(
            dt, dx, rightBC, psi_mminus1, H1=HCoeffs.H1, H2=HCoeffs.H2, H3=HCoeffs.H3, H4=HCoeffs.H4,
            H6=HCoeffs.H6, H7=HCoeffs.H7))

    def solve_matrix_eqn(self, matrixEqn):
        return {self.label: HToMatrixFD.solve(matrixEqn.A, matrixEqn.B, matrixEqn.C, matrixEqn.f)}

    def calculate_residual(self, matrixEqn, profiles):
        resid, _ = _ops.residual(matrixEqn.A, matrixEqn.B, matrixEqn.C, matrixEqn.f, profiles[self.label])
        return {self.label: resid}


class JKCoefficients(object):
    """The H coefficients of two coupled fields side by side: J* of the first, K* of the second."""
    _names = ('1', '2', '3', '4', '6', '7', '8')

    def __init__(self, J1=None, J2=None, J3=None, J4=None, J6=None, J7=None, J8=None,
                 K1=None, K2=None, K3=None, K4=None, K6=None, K7=None, K8=None):
        self.J1, self.J2, self.J3, self.J4, self.J6, self.J7, self.J8 = J1, J2, J3, J4, J6, J7, J8
        self.K1, self.K2, self.K3, self.K4, self.K6, self.K7, self.K8 = K1, K2, K3, K4, K6, K7, K8

    def as_kwargs(self):
        return {p + k: getattr(self, p + k) for p in 'JK' for k in self._names}


class PairCoupledFieldGroup(FieldGroup):
    """tango/fieldgroups.py:172-256."""

    def __init__(self, label0, label1):
        FieldGroup.__init__(self)
        self.label0 = label0
        self.label1 = label1

    def Hcoeffs_to_matrix_eqn(self, dt, dx, rightBC, psi_mminus1, JKCoeffs):
        return PairCoupledMatrixEqn(*JK_to_matrix_fd.JK_to_matrix(dt, dx, rightBC, psi_mminus1, **JKCoeffs.as_kwargs()))

    def solve_matrix_eqn(self, matrixEqn):
        (U, W) = JK_to_matrix_fd.solve(matrixEqn.UEqnCoeffs, matrixEqn.WEqnCoeffs)
        return {self.label0: U, self.label1: W}

    def calculate_residual(self, matrixEqn, profiles):
        Ue, We = matrixEqn.UEqnCoeffs, matrixEqn.WEqnCoeffs
        U, W = profiles[self.label0], profiles[self.label1]
        residU, _ = _ops.residual(Ue.A, Ue.B, Ue.C, Ue.D, U, Ue.G, W)
        residW, _ = _ops.residual(We.A, We.B, We.C, We.D, W, We.G, U)
        return {self.label0: residU, self.label1: residW}


def Hcoeffs_to_JKcoeffs(field1HCoeffs, field2HCoeffs):
    kw = {}
    for k in JKCoefficients._names:
        kw['J' + k] = getattr(field1HCoeffs, 'H' + k)
        kw['K' + k] = getattr(field2HCoeffs, 'H' + k)
    return JKCoefficients(**kw)


def fields_to_fieldgroups(fields, HCoeffsAllFields):
    """Groups in order of first appearance; a coupled pair is emitted where its first member appears
    (tango/fieldgroups.py:275-331)."""
    groups = []
    done = set()
    for field in fields:
        if field.label in done:
            continue
        if field.coupledTo is None:
            fg = UncoupledFieldGroup(field.label)
            fg.HCoeffs = HCoeffsAllFields[field.label]
            fg.rightBC = field.rightBC
            fg.psi_mminus1 = field.profile_mminus1
            done.add(field.label)
        elif field.coupledTo not in done:
            partner = multifield.get_field_by_label(fields, field.coupledTo)
            fg = PairCoupledFieldGroup(field.label, partner.label)
            fg.HCoeffs = Hcoeffs_to_JKcoeffs(HCoeffsAllFields[field.label], HCoeffsAllFields[partner.label])
            fg.rightBC = (field.rightBC, partner.rightBC)
            fg.psi_mminus1 = (field.profile_mminus1, partner.profile_mminus1)
            done.update((field.label, partner.label))
        else:
            continue
        groups.append(fg)
    return groups


def fieldgroups_to_profiles(fieldGroups):
    profiles = {}
    for fg in fieldGroups:
        profiles.update(fg.profileSolution)
    return profiles
