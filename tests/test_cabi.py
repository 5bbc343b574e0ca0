"""The C-ABI library: every symbol include/mm3dgs.h declares is exported, sizes are sane.  No compute (no GPU here)."""
import os
import re

import pytest

from mm3dgs_slam_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, "include", "mm3dgs.h")).read()
    return sorted(set(re.findall(r"\b(mm3dgs_[a-z_]+)\s*\(", h)))


def test_header_and_binding_agree():
    assert _declared() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    import re
    abi = int(re.search(r"#define\s+MM3DGS_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "mm3dgs.h")).read()).group(1))
    assert lib.mm3dgs_version() == abi >= 204      # the header documents the version the library reports
    assert lib.mm3dgs_geom_bytes(1000) >= 1000 * 48
    assert lib.mm3dgs_image_bytes(480, 640) >= 480 * 640 * 8
    assert lib.mm3dgs_binning_bytes(1000) >= 1000 * 41
    assert lib.mm3dgs_backward_scratch_bytes(1000, 5000) >= 5000 * 4 * 48
    assert lib.mm3dgs_last_error() is not None


def test_product_path_refuses_cpu_tensors():
    import torch
    from mm3dgs_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    z = torch.zeros(4, 3)
    rs = GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, False)
    with pytest.raises(RuntimeError, match="no CPU rasterizer"):
        GaussianRasterizer(rs)(means3D=z, means2D=z, opacities=z[:, :1], colors_precomp=z, scales=z, rotations=torch.zeros(4, 4))


def test_drop_in_module_name():
    import diff_gaussian_rasterization as d
    assert d.GaussianRasterizationSettings._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                                                       "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug")


def test_struct_layouts_of_the_binding_match_the_header(tmp_path):
    """Every ctypes.Structure of the binding has the size and the field offsets the C compiler gives the struct of the same name in
    include/mm3dgs.h (gcc on a generated probe; the header is plain C)."""
    import ctypes as C
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    structs = {n: c for n, c in vars(_lib).items() if isinstance(c, type) and issubclass(c, C.Structure) and n.startswith("Mm3dgs")}
    assert len(structs) >= 10
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mm3dgs.h"', 'int main(void) {']
    for n, c in sorted(structs.items()):
        lines.append(f'  printf("{n} %zu\\n", sizeof({n}));')
        for f in c._fields_:
            lines.append(f'  printf("{n}.{f[0]} %zu\\n", offsetof({n}, {f[0]}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])     # (the header must be clean C)
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for n, c in structs.items():
        assert int(got[n]) == C.sizeof(c), (n, got[n], C.sizeof(c))
        for f in c._fields_:
            assert int(got[f"{n}.{f[0]}"]) == getattr(c, f[0]).offset, (n, f[0])


def test_flag_constants_of_the_binding_equal_the_headers_defines():
    import re
    from mm3dgs_slam_amd import _lib
    text = open(os.path.join(ROOT, "include", "mm3dgs.h")).read()
    defines = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+MM3DGS_FWD_(\w+)\s+(\d+)", text)}
    assert defines == {"STATE_CLEAN": _lib.FWD_STATE_CLEAN, "SHORT_LISTS": _lib.FWD_SHORT_LISTS, "DIRECT_BINS": _lib.FWD_DIRECT_BINS,
                       "KEEP_TILE_ORDER": _lib.FWD_KEEP_TILE_ORDER, "PROJECTED": _lib.FWD_PROJECTED}, defines
    assert len(set(defines.values())) == len(defines) and all(v & (v - 1) == 0 for v in defines.values())     # distinct single bits


def test_product_sources_carry_no_timing_probe():
    """The MM3DGS_EXP probes (launches with a phase removed: invalid results) exist only behind -DMM3DGS_PROBES: the product build neither
    reads the environment variable nor tests a probe word inside a kernel."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, "mm3dgs_slam_amd", "csrc")
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        text = open(f).read()
        assert "cam.exp" not in text, f
        if "MM3DGS_EXP" in text:      # only inside the #ifdef MM3DGS_PROBES block of mm3dgs_common.h (and comments that name the variable)
            code = [l for l in text.splitlines() if "MM3DGS_EXP" in l and not l.lstrip().startswith("//") and "//" not in l.split("MM3DGS_EXP")[0]]
            assert all("env_flag" in l for l in code) and f.endswith("mm3dgs_common.h"), (f, code)
    if os.path.exists(_lib.LIB_PATH):
        out = subprocess.run(["strings", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
        assert "MM3DGS_EXP" not in out
