"""The C-ABI library loads on a machine without a GPU and exports every
symbol include/ipx.h declares (no compute calls here)."""
import ctypes
import os
import re

from ipsolver import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ipx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ipx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_hip.LIB_PATH)
    names = declared_symbols()
    assert len(names) > 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_covers_the_header():
    bound = set(_hip.exported_symbols()) | set(_hip._EXTRA_ARGTYPES)
    assert set(declared_symbols()) <= bound, sorted(set(declared_symbols()) - bound)


def test_host_only_entry_points():
    lib = _hip.load()
    assert lib.ipx_version().decode().startswith("ipx")
    assert lib.ipx_banded_kmax() >= 1
    assert lib.ipx_cg_state_size() >= 14
    assert lib.ipx_dense_padded(33) == 64
    import numpy as np
    rowptr = np.array([0, 3, 3, 5000, 5001], dtype=np.int32)
    tiles = np.zeros(8, dtype=np.int32)
    nt = lib.ipx_csr_tiles_host(4, rowptr.ctypes.data_as(ctypes.c_void_p), 2048, 1024,
                                tiles.ctypes.data_as(ctypes.c_void_p), 8)
    assert nt == 3 and list(tiles[:4]) == [0, 2, 3, 4]      # the long row gets its own tile
    assert list(tiles[4:8]) == [0, 3, 5000, 5001]           # rowptr at the tile boundaries


def test_product_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ipsolver import device
    with pytest.raises(_hip.IpxError, match="GPU-only"):
        device.DVec.from_host([1.0, 2.0])
    import ipsolver
    import problems
    p = problems.Maratos()
    with pytest.raises(_hip.IpxError):
        ipsolver.minimize_constrained(p.fun, p.x0, p.grad, p.hess, p.constraints(ipsolver))


def test_header_is_plain_c_and_structs_match_the_bindings(tmp_path):
    """include/ipx.h is the C ABI: it must compile as C99 on its own, and the argument blocks
    mirrored with ctypes must have the layout the compiler gives them."""
    import shutil
    import subprocess
    import pytest
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    from ipsolver.boxschur import BoxSchurArgs
    from ipsolver.cg_fused import CgArgs
    from ipsolver.projector import PcgArgs
    from ipsolver.sharded import Shard2Ext
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ipx.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(ipx_cg_args), sizeof(ipx_boxschur_args), sizeof(ipx_shard2_ext), '
                   'offsetof(ipx_cg_args, state), offsetof(ipx_shard2_ext, own_hi), '
                   'sizeof(ipx_pcg_args), offsetof(ipx_shard2_ext, peer), '
                   'offsetof(ipx_cg_args, no_radius)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True,
                                          text=True).stdout.split()]
    assert got == [ctypes.sizeof(CgArgs), ctypes.sizeof(BoxSchurArgs), ctypes.sizeof(Shard2Ext),
                   CgArgs.state.offset, Shard2Ext.own_hi.offset, ctypes.sizeof(PcgArgs),
                   Shard2Ext.peer.offset, CgArgs.no_radius.offset]


def test_reduction_grid_and_fold_descriptor_match_the_library():
    """The host computes the number of partials a one-launch reduction leaves (device.py
    _reduce_grid) and mirrors ipx_fold_desc: both must agree with the library / the header."""
    import shutil
    import subprocess
    import tempfile
    from ipsolver import _hip, device
    lib = _hip.load()
    for n in (1, 1023, 1024, 1025, 123456, 10 ** 6, 10 ** 6 + 1, 16 * 10 ** 6, 10 ** 9):
        assert int(lib.ipx_reduce_grid(n)) == device._reduce_grid(n), n
    assert int(lib.ipx_reduce_grid(10 ** 12)) == _hip.VEC_GRID_CAP
    if shutil.which("gcc") is None:
        return
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "fd.c")
        with open(src, "w") as f:
            f.write('#include <stdio.h>\n#include <stddef.h>\n#include "ipx.h"\n'
                    'int main(void) { printf("%zu %zu %zu %d\\n", sizeof(ipx_fold_desc), '
                    'offsetof(ipx_fold_desc, count), offsetof(ipx_fold_desc, op), IPX_FOLD_MAX); '
                    'return 0; }\n')
        exe = os.path.join(tmp, "fd")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        src, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], check=True, capture_output=True,
                                              text=True).stdout.split()]
    D = device._FoldDesc
    assert got == [ctypes.sizeof(D), D.count.offset, D.op.offset, device.FOLD_MAX]
