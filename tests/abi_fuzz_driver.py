"""Driver of tests/test_abi_asan.py: runs INSIDE a subprocess that has the AddressSanitizer runtime preloaded and no visible
GPU, loads the ASAN build of the C-ABI library and walks the host side of every entry point -- argument checks, tile / split-K
planners, workspace-size queries, launch-geometry arithmetic.  No kernel can run (no device): a call that gets past its
argument checks ends in a HIP launch error code.  Any memory error, integer division by zero or abort in host code kills the
process with an ASAN report, which the test asserts does not happen.

usage: python abi_fuzz_driver.py <library.so> <seed>"""
import ctypes
import random
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec._hip import binding as B  # noqa: E402  (header parser + struct mirrors; does not load the product library)

HOST_STRUCTS = {"ebn_encoder_dims": B.EncoderDims, "ebn_encoder_params": B.EncoderParams, "ebn_encoder_acts": B.EncoderActs,
                "ebn_encoder_grads": B.EncoderGrads, "ebn_encoder_scratch": B.EncoderScratch, "ebn_finish_job": B.FinishJob * B.FINISH_MAX_JOBS,
                "ebn_dvn_args": B.DvnArgs, "ebn_tn_problem": B.TnProblem * B.TN_GROUP_MAX, "ebn_dvn_finale": B.DvnFinale, "ebn_adam_flat": B.AdamFlat}
FAKE_DEV = 0x7E0000000000  # a 16-byte-aligned address no host mapping uses: device pointers are never dereferenced on the host
SIZES = [0, 1, 2, 3, 5, 7, 16, 20, 30, 31, 32, 33, 50, 63, 64, 65, 100, 127, 128, 200, 255, 256, 257, 300, 400, 512, 768, 1000, 1024, 1200,
         4096, 24000, 32000, 52800, 250002, 1 << 20, (1 << 24) + 1, (1 << 31) - 1, 1 << 31, (1 << 31) + 7, 1 << 33, 1 << 40, (1 << 62) + 3]


def prototypes():
    text = re.sub(r"/\*.*?\*/", "", B.header_path().read_text(), flags=re.S)
    for ret, name, args in B._PROTO.findall(text):
        yield name, ret, [a.strip() for a in args.replace("\n", " ").split(",") if a.strip() not in ("", "void")]


def value(decl, rng, mode, keep):
    """one argument: mode "null" -> NULL / 0; "neg" -> -1; "rand" -> fake device pointers, real host structs, sizes from SIZES"""
    base = decl.rsplit(" ", 1)[0].replace("const", "").strip()
    if "*" in decl or decl.startswith("ebn_stream_t"):
        if mode != "rand" or decl.startswith("ebn_stream_t"):
            return None
        stype = next((t for n, t in HOST_STRUCTS.items() if n in decl), None)
        if stype is not None:  # host structs the entry point reads: real memory, random contents
            s = stype()
            if isinstance(s, ctypes.Array):  # a host ARRAY of structs (ebn_finish_job jobs[]): random contents in every element
                for el in s:
                    for fname, ftype in el._fields_:
                        if ftype is ctypes.c_void_p:
                            setattr(el, fname, None if rng.random() < 0.1 else FAKE_DEV + 4096 * rng.randrange(1 << 16))
                        elif ftype is ctypes.c_float:
                            setattr(el, fname, rng.choice([0.0, 1.0, -1.0]))
                        else:
                            bits = 63 if ftype is ctypes.c_int64 else 31
                            setattr(el, fname, rng.choice([v for v in SIZES[:30] + [-1, 0, 1, 2, 3] if v < (1 << bits)]))
                keep.append(s)
                return ctypes.cast(s, ctypes.c_void_p)
            for fname, ftype in s._fields_:
                if isinstance(ftype, type) and issubclass(ftype, ctypes.Array):  # fixed arrays inside a struct (ebn_dvn_args)
                    arr = getattr(s, fname)
                    for i in range(len(arr)):
                        if ftype._type_ is ctypes.c_void_p:
                            arr[i] = None if rng.random() < 0.05 else FAKE_DEV + 4096 * rng.randrange(1 << 16)
                        else:
                            arr[i] = rng.choice([v for v in SIZES + [-1] if v < (1 << 31)])
                elif ftype is ctypes.c_void_p:
                    setattr(s, fname, None if rng.random() < 0.05 else FAKE_DEV + 4096 * rng.randrange(1 << 16))
                elif ftype is ctypes.c_float:
                    setattr(s, fname, rng.choice([0.0, 0.2, -1.0, 1.5]))
                else:
                    bits = 63 if ftype is ctypes.c_int64 else 31
                    setattr(s, fname, rng.choice([v for v in SIZES + [-1] if v < (1 << bits)]))
            keep.append(s)
            return ctypes.cast(ctypes.pointer(s), ctypes.c_void_p)
        if decl.startswith("int32_t*") and decl.split()[-1] in ("bm", "bn", "splits", "n_parts"):  # host out-parameters of ebn_gemm_plan
            v = ctypes.c_int32()
            keep.append(v)
            return ctypes.cast(ctypes.pointer(v), ctypes.c_void_p)
        u = rng.random()  # mostly plausible pointers, so that the calls get past the NULL checks into the planners
        return ctypes.c_void_p(0 if u < 0.05 else (FAKE_DEV + 4 if u < 0.10 else FAKE_DEV + 4096 * rng.randrange(1 << 16)))  # (+4: misaligned)
    if base in ("float", "double"):
        return {"null": 0.0, "neg": -1.0}.get(mode, rng.choice([0.0, 1.0, -1.0, 0.2, 1e30]))
    bits = 63 if base == "int64_t" else 31
    if mode == "null":
        return 0
    if mode == "neg":
        return -1
    pool = [v for v in SIZES + [-1, -7] if v < (1 << bits)]
    return rng.choice(pool[: len(pool) * 2 // 3]) if rng.random() < 0.7 else rng.choice(pool)  # mostly sizes a real problem has


def main():
    lib = ctypes.CDLL(sys.argv[1])
    rng = random.Random(int(sys.argv[2]))
    decl = B.declared_functions()
    n_calls, codes = 0, {}
    for name, ret, args in prototypes():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = decl[name]
        if ret == "const char*":
            for c in (-5, -1, 0, 1, 100, 98765):
                assert fn(c) is not None
            continue
        for mode, reps in (("null", 1), ("neg", 1), ("rand", 300)):
            for _ in range(reps):
                keep = []
                rc = fn(*[value(a, rng, mode, keep) for a in args])
                n_calls += 1
                codes[rc if ret == "int" else "query"] = codes.get(rc if ret == "int" else "query", 0) + 1
                if ret == "int64_t":
                    assert rc >= 0, (name, rc)  # a size query never reports a negative size
    print(f"ABI_FUZZ_OK calls={n_calls} codes={sorted(codes.items(), key=str)}")


if __name__ == "__main__":
    main()
