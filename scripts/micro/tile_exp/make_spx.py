"""Writes spx.h = csrc/leafnet_sp.h with early exits (-DX_DBG=n) for sp_exp.hip: 1 after the stem, 2 after the trunk, 3 after the head
1x1 convolutions, 4 after the value head's extra convolution + pool, 5 after the policy head's convolutions and logits, 0 = the whole tile.
Generated (not committed): the product header is the source, the experiment only adds exits."""
import os
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "..", "..", "..", "alphazero-pybind11_amd", "csrc", "leafnet_sp.h")).read()
rules = [
    ('#include "leafnet_c4.h"',
     '#include "../../../alphazero-pybind11_amd/csrc/leafnet_c4.h"\n#ifndef X_DBG\n#define X_DBG 0\n#endif\n'
     '#define X_EXIT(n, acc) do { if (X_DBG == n) { if (acc[0][0][0] == 12345.678f) vpool[0] = acc[1][1][1]; return; } } while (0)'),
    ('  int slot = nd.C_in <= 8 ? conv(kstem8, s, 0) : conv(k3x3, s, 0);\n',
     '  int slot = nd.C_in <= 8 ? conv(kstem8, s, 0) : conv(k3x3, s, 0);\n  X_EXIT(1, s);\n'),
    ('  // ---- head 1x1 convs over the raw stream: hv (value rows), hp (policy rows)',
     '  X_EXIT(2, s);\n  // ---- head 1x1 convs over the raw stream: hv (value rows), hp (policy rows)'),
    ('  slot = conv(k1x1, hp, slot);\n\n  // average pool', '  slot = conv(k1x1, hp, slot);\n  X_EXIT(3, hp);\n\n  // average pool'),
    ('  pool(hv, vp);\n  rezero_cells();',
     '  pool(hv, vp);\n  rezero_cells();\n  if (X_DBG == 4) { if (vp[0][0] == 12345.678f) vpool[0] = vp[1][0] + hp[0][0][0]; return; }'),
    ('  put_pooled(vpool, vp);\n  if (nd.num_global > 0) {', '  put_pooled(vpool, vp);\n  if (X_DBG == 5) return;\n  if (nd.num_global > 0) {'),
]
for old, new in rules:
    assert src.count(old) == 1, old
    src = src.replace(old, new)
open(os.path.join(HERE, "spx.h"), "w").write(src)
print("wrote spx.h")
