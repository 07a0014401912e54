"""The C-ABI library loads and exports every symbol include/hificar.h declares; host-only entry points
(create / set_weight / workspace_bytes / macs / error reporting) behave.  No compute calls: no GPU here."""

import ctypes
import os
import re

import numpy as np
import pytest

from conftest import E2W_PARAMS, REPO
from articulatory_amd import _native


@pytest.fixture(scope="module")
def lib():
    return _native.load_library()


def _full_params(**over):
    p = dict(E2W_PARAMS, use_tanh=True)
    p.update(over)
    return p


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(REPO, "include", "hificar.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(hificar_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_native.SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert b"gfx950" in lib.hificar_version()


def test_config_struct_matches_header_layout(tmp_path):
    # 5 scalars + 8 + 8 + 1 + 4 + 4 + 16 + 2 ints + float + 6 ints + 7 conditioning ints = 63 x 4 bytes
    assert ctypes.sizeof(_native.HificarConfig) == 4 * (5 + 8 + 8 + 1 + 4 + 4 + 16 + 2 + 1 + 6 + 7)
    assert ctypes.sizeof(_native.HificarKernelStat) == 96 + 8 + 3 * 8
    # and against the C compiler's view of include/hificar.h: size and the offset of every field
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    fields = [f[0] for f in _native.HificarConfig._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hificar.h"\nint main(void) {\n'
                   '  printf("%zu\\n", sizeof(hificar_config));\n' +
                   "".join(f'  printf("%zu\\n", offsetof(hificar_config, {f}));\n' for f in fields) +
                   '  printf("%zu\\n", sizeof(hificar_kernel_stat));\n  return 0;\n}\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [ctypes.sizeof(_native.HificarConfig)] + [getattr(_native.HificarConfig, f).offset for f in fields] + \
           [ctypes.sizeof(_native.HificarKernelStat)]
    assert got == want


def test_training_structs_match_header_layout(tmp_path):
    """The discriminator / loss / mel structs of the training entry points: ctypes mirrors vs the C header, field by field."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    pairs = [("hificar_gblock_config", _native.HificarGBlockConfig),
             ("hificar_disc_config", _native.HificarDiscConfig), ("hificar_disc_output", _native.HificarDiscOutput),
             ("hificar_gan_loss_config", _native.HificarGanLossConfig), ("hificar_mel_config", _native.HificarMelConfig)]
    body, want = "", []
    for cname, cls in pairs:
        body += f'  printf("%zu\\n", sizeof({cname}));\n'
        want.append(ctypes.sizeof(cls))
        for f in cls._fields_:
            body += f'  printf("%zu\\n", offsetof({cname}, {f[0]}));\n'
            want.append(getattr(cls, f[0]).offset)
    src = tmp_path / "layout2.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hificar.h"\nint main(void) {\n' + body + "  return 0;\n}\n")
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    exe = tmp_path / "layout2"
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got == want
    assert _native.DISC_MAX_SUBS == 8 and _native.DISC_MAX_LAYERS == 12  # HIFICAR_DISC_MAX_* of the header
    hdr = open(os.path.join(inc, "hificar.h")).read()
    assert "#define HIFICAR_DISC_MAX_SUBS 8" in hdr and "#define HIFICAR_DISC_MAX_LAYERS 12" in hdr
    assert _native.MAX_GBLOCKS == 10 and "#define HIFICAR_MAX_GBLOCKS 10" in hdr


def test_create_macs_workspace(lib):
    cfg = _native.make_config(_full_params(), _native.PREC_F32)
    h = ctypes.c_void_p()
    assert lib.hificar_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    try:
        mlp = 512 * 256 + 3 * 256 * 256 + 256 * 128
        # 117 668 864 conv MACs per frame (HiFi-CAR 13-dim) + 360 448 per PastFCEncoder call
        assert lib.hificar_macs(h, 1, 1) == 117668864 + mlp
        assert lib.hificar_macs(h, 64, 25) == 64 * (25 * 117668864 + mlp)
        assert abs(lib.hificar_macs(h, 1, 25) / 2000 - 1471041) < 1.0  # MAC per output sample at chunk 25
        ws = lib.hificar_workspace_bytes(h, 64, 25)
        assert 100e6 < ws < 200e6
        assert lib.hificar_workspace_bytes(h, 0, 25) == 0
        # forward before finalize is a state error, not a crash
        rc = lib.hificar_forward(h, 1, 1, 1, 1, 1, 1, 0, None)
        assert rc == -2 and b"finalize" in lib.hificar_last_error()
        rc = lib.hificar_finalize(h)
        assert rc == -2 and b"Missing key" in lib.hificar_last_error()
    finally:
        lib.hificar_destroy(h)


def test_set_weight_validation(lib):
    cfg = _native.make_config(_full_params(), _native.PREC_F32)
    h = ctypes.c_void_p()
    assert lib.hificar_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    try:
        w = np.zeros((512, 141, 7), dtype=np.float32)
        shp = (ctypes.c_int64 * 3)(512, 141, 7)
        assert lib.hificar_set_weight(h, b"input_conv.weight", w.ctypes.data, shp, 3) == 0
        bad = (ctypes.c_int64 * 3)(512, 140, 7)
        assert lib.hificar_set_weight(h, b"input_conv.weight", w.ctypes.data, bad, 3) == -1
        assert b"size mismatch for input_conv.weight" in lib.hificar_last_error()
        assert lib.hificar_set_weight(h, b"input_conv.weight_v", w.ctypes.data, shp, 3) == -1
        assert b"unexpected tensor name" in lib.hificar_last_error()
    finally:
        lib.hificar_destroy(h)


@pytest.mark.parametrize("over,msg", [
    (dict(out_channels=4), b"out_channels"),
    (dict(kernel_size=6), b"odd"),
    (dict(upsample_scales=[5, 4, 2, 2], upsample_kernel_sizes=[11, 8, 4, 4]), b"L_out"),
    (dict(channels=8), b"halvings"),
])
def test_create_rejects_unsupported(lib, over, msg):
    cfg = _native.make_config(_full_params(**over), _native.PREC_F32)
    h = ctypes.c_void_p()
    rc = lib.hificar_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == -1
    assert msg in lib.hificar_last_error(), lib.hificar_last_error()


def test_create_accepts_four_blocks_and_single_conv_layers(lib):
    """Up to HIFICAR_MAX_BLOCKS = 4 residual blocks per stage (hifigan.py:134-145) and use_additional_convs = false (residual_block.py:191-205):
    accepted since round 5; a fifth block is refused before the C ABI is reached."""
    for over in (dict(resblock_kernel_sizes=[3, 7, 11, 13], resblock_dilations=[[1]] * 4), dict(use_additional_convs=False)):
        cfg = _native.make_config(_full_params(**over), _native.PREC_F32)
        h = ctypes.c_void_p()
        assert lib.hificar_create(ctypes.byref(cfg), ctypes.byref(h)) == 0, lib.hificar_last_error()
        w = np.zeros((256, 256, 3), dtype=np.float32)
        shp = (ctypes.c_int64 * 3)(256, 256, 3)
        rc = lib.hificar_set_weight(h, b"blocks.0.convs2.0.1.weight", w.ctypes.data, shp, 3)
        assert rc == (0 if "resblock_kernel_sizes" in over else -1)  # no convs2 tensors without the additional convs
        lib.hificar_destroy(h)
    with pytest.raises(ValueError):
        _native.make_config(_full_params(resblock_kernel_sizes=[3, 5, 7, 9, 11], resblock_dilations=[[1]] * 5), _native.PREC_F32)


def test_create_accepts_any_width(lib):
    """The reference builds channels // 2**i wide stages for any `channels` (hifigan.py:108-145); widths below / between MFMA-tile
    multiples are padded to 32 channels internally (no compute call here: creation only needs no GPU)."""
    for channels in (64, 48, 100, 16):
        cfg = _native.make_config(_full_params(channels=channels), _native.PREC_F32)
        h = ctypes.c_void_p()
        assert lib.hificar_create(ctypes.byref(cfg), ctypes.byref(h)) == 0, lib.hificar_last_error()
        lib.hificar_destroy(h)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "libhificar.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.load_library()


def test_header_is_plain_c(tmp_path):
    """include/hificar.h is a C ABI: it must compile as C99 on its own (no C++ / HIP / torch types)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "t.c"
    src.write_text('#include "hificar.h"\nint main(void) { return hificar_version() == 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
