"""Restart within a timestep (tango/restart.py:59-98, state written by tango/handlers.py:211-242).

The state that continues a solve is, per field: the current profile, the three EWMA iterates of the lodestro
method on the turbulence grid (profileEWMATurbGrid, DEWMATurbGrid, cEWMATurbGrid) and profile_mminus1; plus the
last iteration number, t, the timestep number and the set number.  The reference keeps it in the last row of
its HDF5 history file; h5py is not a dependency here, so `write_state` stores the same names in a NumPy `.npz`
(`<basename>_s<setNumber>.npz`).  `read_metadata_from_previousfile` returns the reference's tuple from either
kind of file (HDF5 only when h5py is importable).  Arrays may be (N,) or (profiles, N).
"""
import os

import numpy as np

FIELD_KEYS = ('profile', 'profileEWMATurbGrid', 'DEWMATurbGrid', 'cEWMATurbGrid', 'profile_mminus1')


def write_state(solver, basename='tangodata', setNumber=0):
    """Save what a restart needs from `solver` (after take_timestep returned or between iterations of the generic
    path).  iterationNumber is the index of the last completed iteration, as in the reference's history file.
    Returns the file name."""
    out = {'setNumber': int(setNumber), 't': float(solver.t), 'timestepNumber': int(solver.m),
           'iterationNumber': int(solver.iterationNumber if solver.iterationNumber else (solver.countLastTimestep or 0)) - 1,
           'labels': np.array([f.label for f in solver.fields])}
    for f in solver.fields:
        m = f.lodestroMethod
        vals = (solver.profiles[f.label], m._EWMAProfile._yEWMA_lminus1, m._EWMATurbD._yEWMA_lminus1,
                m._EWMATurbc._yEWMA_lminus1, solver.restart_profile_mminus1[f.label])
        for key, v in zip(FIELD_KEYS, vals):
            if v is None:
                raise ValueError(f'field {f.label}: {key} is not set yet (no iteration has run)')
            out[f'{f.label}/{key}'] = np.asarray(v, dtype=np.float64)
    filename = f'{basename}_s{int(setNumber)}.npz'
    np.savez(filename, **out)
    return filename


def _read_npz(filename):
    with np.load(filename, allow_pickle=False) as z:
        labels = [str(v) for v in z['labels']]
        meta = (int(z['setNumber']), int(z['iterationNumber']), float(z['t']), int(z['timestepNumber']))
        per = {key: {lab: z[f'{lab}/{key}'] for lab in labels} for key in FIELD_KEYS}
    return meta, per


def _read_hdf5(filename):
    import h5py      # optional: only for files written by the reference
    with h5py.File(filename, 'r') as f:
        labels = [k for k in f.keys() if isinstance(f[k], h5py.Group)]
        meta = (int(f.attrs['setNumber']), int(f['iterationNumber'][-1]), float(f.attrs['t']), int(f.attrs['timestepNumber']))
        per = {key: {} for key in FIELD_KEYS}
        for lab in labels:
            grp = f[lab]
            per['profile_mminus1'][lab] = np.array(grp.attrs['profile_mminus1'])
            for key in FIELD_KEYS[:4]:
                per[key][lab] = np.array(grp[key][-1])
    return meta, per


def read_metadata_from_previousfile(filename):
    """(setNumber, startIterationNumber, t, timestepNumber, old_profiles, old_profilesEWMA, old_turb_D_EWMA,
    old_turb_c_EWMA) for the continuation run, as tango/restart.py:59-88."""
    meta, per = (_read_npz if filename.endswith('.npz') else _read_hdf5)(filename)
    old_set, last_iteration, t, timestep = meta
    return (old_set + 1, last_iteration + 1, t, timestep, per['profile'], per['profileEWMATurbGrid'],
            per['DEWMATurbGrid'], per['cEWMATurbGrid'])


def read_profiles_mminus1(filename):
    """profile_mminus1 of every field (the reference stores it as a group attribute, tango/handlers.py:294-309)."""
    return (_read_npz if filename.endswith('.npz') else _read_hdf5)(filename)[1]['profile_mminus1']


def set_ewma_iterates(fields, old_profilesEWMA, old_turb_D_EWMA, old_turb_c_EWMA):
    """tango/restart.py:90-98."""
    for field in fields:
        lab = field.label
        field.lodestroMethod.set_ewma_iterates(old_profilesEWMA[lab], old_turb_D_EWMA[lab], old_turb_c_EWMA[lab])


def find_latest_savefile(basename, extensions=('.npz', '.hdf5')):
    """The save file `<basename>_s<number><ext>` with the largest number in the current directory, or None
    (tango/restart.py:145-163)."""
    start = basename + '_s'
    best = None
    for f in os.listdir(os.getcwd()):
        stem, ext = os.path.splitext(f)
        if ext in extensions and stem.startswith(start) and stem[len(start):].isdigit():
            num = int(stem[len(start):])
            if best is None or num > best[0]:
                best = (num, f)
    return None if best is None else best[1]


def check_if_should_restart(basename=None):
    """tango/restart.py:114-143: the file to restart from, or None."""
    if basename is None:
        files = [f for f in os.listdir(os.getcwd()) if f.endswith(('.npz', '.hdf5')) and '_s' in f]
        if not files:
            return None
        starts = {f[:f.rfind('_')] for f in files}
        if len(starts) != 1:
            raise ValueError('No basename provided, and multiple basenames for save files exist in the current directory '
                             + os.getcwd())
        basename = starts.pop()
    return find_latest_savefile(basename)


def combine_savefiles(basename='tangodata'):
    """Combine the save files `<basename>_s<number>.npz` of the current directory into `<basename>_combined.npz`, in
    increasing set number (tango/restart.py:164-212 does this for its HDF5 history files: the datasets of the later files
    are appended along the first axis).  Every save file here holds one write (the state after its last iteration), so
    the combined file has one row per set: `iterationNumber`, `t`, `timestepNumber`, `setNumber` of shape (sets,) and
    `<label>/<dataset>` of shape (sets,) + the dataset's own shape.  Returns the file name."""
    start = basename + '_s'
    found = []
    for f in os.listdir(os.getcwd()):
        stem, ext = os.path.splitext(f)
        if ext == '.npz' and stem.startswith(start) and stem[len(start):].isdigit():
            found.append((int(stem[len(start):]), f))
    if not found:
        raise ValueError('no files found.')
    rows = []
    for _, f in sorted(found):
        (setNumber, iterationNumber, t, timestepNumber), per = _read_npz(f)
        rows.append((setNumber, iterationNumber, t, timestepNumber, per))
    labels = list(rows[0][4]['profile'])
    if any(list(r[4]['profile']) != labels for r in rows):
        raise ValueError('the save files do not hold the same fields')
    out = {'labels': np.array(labels), 'setNumber': np.array([r[0] for r in rows]), 'iterationNumber': np.array([r[1] for r in rows]),
           't': np.array([r[2] for r in rows]), 'timestepNumber': np.array([r[3] for r in rows])}
    for lab in labels:
        for key in FIELD_KEYS:
            out[f'{lab}/{key}'] = np.stack([np.asarray(r[4][key][lab]) for r in rows])
    filename = basename + '_combined.npz'
    np.savez(filename, **out)
    return filename
