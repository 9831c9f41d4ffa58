import sys, numpy as np
sys.path.insert(0, '.')
import tango_amd as tango
from oracle import tango_oracle as O
g = np.load('tests/golden/g3_pair.npz')
r = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
z = np.zeros_like(g['x'])
kw = {f'J{i}': g[f'J{i}'] for i in (1, 2, 3, 6, 7, 8)}
kw.update({f'K{i}': g[f'K{i}'] for i in (1, 2, 3, 6, 7, 8)})
Ue, We = tango.JK_to_matrix_fd.JK_to_matrix(float(g['dt']), float(g['dx']), (float(g['UL']), float(g['WL'])), (z, z), **kw)
Ug = tuple(g[k] for k in ('AU', 'BU', 'CU', 'DU', 'GW'))
Wg = tuple(g[k] for k in ('AW', 'BW', 'CW', 'DW', 'GU'))
for a, b, k in zip(Ue + We, Ug + Wg, 'AU BU CU DU GW AW BW CW DW GU'.split()):
    print(k, 'max abs diff', np.max(np.abs(a - b)), 'max', np.max(np.abs(b)))
from types import SimpleNamespace as NS
mk = lambda t: NS(A=t[0], B=t[1], C=t[2], D=t[3], G=t[4])
U1, W1 = tango.JK_to_matrix_fd.solve(mk(Ug), mk(Wg))
print('gpu solve of golden matrices', r(U1, g['U']), r(W1, g['W']))
U2, W2 = O.pair_solve(Ug, Wg)
try:
    print('dgbsv of gpu matrices', r(U2, g['U']), r(W2, g['W']))
except Exception as e:
    print('oracle call failed', e)
U3, W3 = tango.JK_to_matrix_fd.solve(Ue, We)
print('gpu solve of gpu matrices', r(U3, g['U']), r(W3, g['W']))
print('where', np.argmax(np.abs(U3 - g['U'])), np.argmax(np.abs(U1 - g['U'])))

e = np.abs(U3 - g['U']) / np.max(np.abs(g['U']))
print('err by position (every 25):', np.array2string(e[::25], precision=2))
sys.path.insert(0, '/root/repo/scripts')
