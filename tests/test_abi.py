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
