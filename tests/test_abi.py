"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and
exports exactly the entry points include/ebnerd_hip.h declares (no compute calls here)."""
import ctypes
import subprocess

from ebrec import _hip


def test_library_is_built_in_tree():
    assert _hip.library_path().exists(), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported_and_typed():
    handle = _hip.lib()
    decl = _hip.declared_functions()
    assert len(decl) >= 25
    for name, (res, argt) in decl.items():
        fn = getattr(handle, name)
        assert fn.restype is res and list(fn.argtypes) == argt, name


def test_no_undeclared_ebn_symbols_leak():
    out = subprocess.run(["nm", "-D", "--defined-only", str(_hip.library_path())], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("ebn_")}
    assert exported == set(_hip.declared_functions()), exported ^ set(_hip.declared_functions())


def test_error_strings_and_version():
    handle = _hip.lib()
    assert handle.ebn_abi_version() == 1
    assert b"bad argument" in handle.ebn_error_string(-1)
    assert b"not supported" in handle.ebn_error_string(-2)


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(_hip.StepState) == 64
    assert ctypes.sizeof(_hip.EncoderDims) == 40
    assert ctypes.sizeof(_hip.EncoderScratch) == 48


def _plan(M, N, K, ws=1 << 60):
    bm, bn, sp = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert _hip.lib().ebn_gemm_plan(M, N, K, ws, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sp)) == 0
    return bm.value, bn.value, sp.value


def test_gemm_planner_is_a_pure_host_query_with_sane_plans():
    """ebn_gemm_plan / ebn_gemm_workspace_floats never touch the device: tile in the four families, split-K only
    with a workspace that can hold every partial, and the shapes DESIGN.md quotes pick what it says they pick."""
    wsf = _hip.lib().ebn_gemm_workspace_floats
    for (M, N, K) in [(1, 1, 1), (5, 7, 3), (640, 1200, 400), (400, 200, 24000), (24000, 1200, 1024), (1024, 1200, 24000),
                      (24000, 400, 200), (4096, 4096, 4096), (300, 1200, 24000), (97, 4099, 17)]:
        bm, bn, sp = _plan(M, N, K)
        assert (bm, bn) in ((128, 128), (64, 64), (256, 64), (128, 64), (32, 32)) and 1 <= sp <= 64
        assert sp == 1 or bm != 32, "the 32x32 small-output kernel never splits K"
        assert int(wsf(M, N, K)) == (sp * M * N if sp > 1 else 0)
        assert _plan(M, N, K, 0)[2] == 1, "no workspace, no split"
        if sp > 1:  # a workspace one float short of `sp` partials must make the planner settle for fewer splits
            assert _plan(M, N, K, sp * M * N - 1)[2] < sp
    assert _plan(24000, 1200, 1024) == (256, 64, 1)  # Q|K|V projection: 1786 workgroups = 6.98 turns, N pads to 1216
    assert _plan(1024, 1200, 24000)[:2] == (256, 64) and _plan(1024, 1200, 24000)[2] > 1  # its weight gradient: split-K
    assert _plan(640, 1200, 400)[2] == 1  # user-encoder projection: one launch, a reduce would cost more than it saves
    assert _plan(800, 512, 768) == (32, 32, 1)  # DocVec Dense layer: 400 small workgroups instead of split-K + reduce
    # the measured picks of profiles/r02_gemm_tuning.md: the small-output kernel wherever it applies (user encoder), the
    # AttLayer2 shapes, the skinny weight gradient of a 300-wide table with a long K range (six workgroups per CU)
    assert _plan(640, 1200, 400) == (32, 32, 1) and _plan(640, 400, 1200) == (32, 32, 1) and _plan(400, 1200, 640) == (32, 32, 1)
    assert _plan(24000, 200, 400) == (128, 64, 1) and _plan(24000, 400, 200) == (128, 128, 1) and _plan(400, 200, 24000) == (64, 64, 18)
    assert _plan(300, 1200, 24000) == (64, 64, 16) and _plan(24000, 300, 1200) == (256, 64, 1)
    assert _hip.lib().ebn_gemm_plan(-1, 1, 1, 0, None, None, None) == -1


def test_gemm_planner_small_tile_choice_is_bounded_to_its_measured_envelope():
    """Round-2 ADVICE: the 32x32 small-output kernel was returned unconditionally whenever tiles64 <= 256 and K <= 4096.
    It is now taken outright only inside the envelope it was measured on (K <= 1536, at most three workgroups on the busiest
    CU); outside it competes with the big tiles on modelled cost.  Edges: K = 1536 / 1537 / 4096 / 4097, tiles64 = 256 / 257."""
    # inside the envelope: every step shape of c2 / c3 / c4 that used it keeps it
    for shape in [(640, 1200, 400), (640, 400, 1200), (400, 1200, 640), (800, 512, 768), (640, 256, 200), (1600, 400, 1200)]:
        assert _plan(*shape) == (32, 32, 1), shape
    assert _plan(640, 1200, 1536) == (32, 32, 1)  # K edge of the envelope (W = 3)
    # a fourth workgroup per CU at a long K is outside it: 1024 x 1024 (tiles64 = 256, W = 4) x 4096 -- whatever the model picks
    # must be a valid plan, and a long, well-filled K range must not stay on the never-splitting small kernel
    for shape in [(1024, 1024, 4096), (1024, 1024, 1537), (640, 1200, 1537), (640, 1200, 4096), (640, 1200, 4097), (1024, 1088, 1024)]:
        bm, bn, sp = _plan(*shape)
        assert (bm, bn) in ((128, 128), (64, 64), (256, 64), (128, 64), (32, 32)) and 1 <= sp <= 64 and (sp == 1 or bm != 32), shape
    assert _plan(640, 1200, 4097)[0] != 32 and _plan(1024, 1088, 1024)[0] != 32  # K > 4096 / tiles64 = 272: never the small kernel
    assert _plan(1024, 1024, 4096)[0] != 32  # modelled: 32 slabs x 4 workgroups per CU loses to 128x128 tiles with split-K


def test_integration_doc_lists_exactly_the_exported_entry_points():
    """INTEGRATION.md's table of entry points by reference symbol stays in sync with the header."""
    import re
    from pathlib import Path

    text = (Path(__file__).resolve().parents[1] / "INTEGRATION.md").read_text()
    mentioned = set(re.findall(r"`(ebn_[a-z0-9_]+)`", text))
    assert mentioned == set(_hip.declared_functions()), mentioned ^ set(_hip.declared_functions())
