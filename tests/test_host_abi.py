"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol declared in
include/aindex_hip.h, exact-modulo / codec self-tests, MWHC builder bit-identity with the reference's
compute_mphf_seq (golden .pf), record normalisation. No GPU compute calls."""
import hashlib
import json
import os

import ctypes as C
import numpy as np
import pytest

from aindex_amd import _lib, builder, synth
import oracle_lib as O


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _lib.header_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), f"{name} declared in aindex_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert b"gfx950" in L.aix_version()


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.AixError):
        _lib.device_count()
    h = C.c_void_p()
    st = _lib.lib().aix_index_open_23(b"/nonexistent.pf", b"/x", b"/y", 0, C.byref(h))
    assert st != 0


def test_fastmod_exact():
    L = _lib.lib()
    rng = np.random.default_rng(5)
    ds = [1, 2, 3, 5, 7, 255, 256, 257, 65535, 65536, 1664, 27514635, 20500001, 2 ** 31 - 1, 2 ** 31, 2 ** 31 + 1,
          2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 40 + 12345, 2 ** 63 + 1] + [int(x) for x in rng.integers(1, 2 ** 32, 200)]
    hs = [0, 1, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 64 - 1, 2 ** 64 - 2, 2 ** 63] + [int(x) for x in rng.integers(0, 2 ** 64, 400, dtype=np.uint64)]
    for d in ds:
        for h in hs + [d - 1, d, d + 1, 2 * d, (2 ** 64 - 1) // d * d, ((2 ** 64 - 1) // d * d - 1) % 2 ** 64]:
            h %= 2 ** 64
            assert L.aix_selftest_mod(h, d) == h % d, (h, d)


def test_revcomp_matches_reference_codec(gold):
    L = _lib.lib()
    k = json.load(open(os.path.join(gold, "codec_kat.json")))
    for e, r in zip(k["enc23"], k["rev23"]):
        assert L.aix_selftest_revcomp(e, 23) == r
    for e, r in zip(k["enc13"], k["rev13"]):
        assert L.aix_selftest_revcomp(e, 13) == r


def test_builder_bit_identical_small23(small23_prefix):
    rows = [ln.split("\t")[0] for ln in open(small23_prefix + ".dat").read().split("\n") if ln]
    want = open(small23_prefix + ".pf", "rb").read()
    assert builder.build_pf(rows) == want
    assert builder.build_pf_fixed("".join(rows).encode(), 23) == want


def test_builder_rejects_duplicates():
    with pytest.raises(_lib.AixError):
        builder.build_pf(["ACGTACGTACGTACGTACGTACG"] * 4)


def test_builder_random_sets_are_minimal_perfect():
    for n, seed in ((1, 1), (3, 2), (37, 3), (5000, 4)):   # n=2 gives hash domain 1: never peelable, also in the reference
        kmers = np.unique(synth.random_kmers_ascii(seed, n, 23), axis=0)
        pf = builder.build_pf_fixed(kmers, 23)
        path = f"/tmp/aix_t_{n}.pf"
        open(path, "wb").write(pf)
        m = O.OracleMphf(path)
        got = sorted(m.lookup(bytes(k)) for k in kmers)
        assert got == list(range(kmers.shape[0]))


@pytest.mark.parametrize("name", ["refdata_test.fasta", "refdata_test_se.fastq", "refdata_test_reads.txt", "synth.fa", "synth.fq", "synth.txt"])
def test_normalize_then_plain_rule_equals_reader(gold, name):
    """aix_normalize_reads + the PLAIN window rule == the reference's per-format readers (via the oracle)."""
    from pf13 import pf13_path
    buf = open(os.path.join(gold, "count13", name), "rb").read()
    a = np.frombuffer(buf, dtype=np.uint8)
    out = np.empty(a.shape[0] + 2, dtype=np.uint8)
    n = C.c_uint64()
    _lib.check(_lib.lib().aix_normalize_reads(a.ctypes.data_as(C.c_void_p), a.shape[0], -1, 0, out.ctypes.data_as(C.c_void_p), C.byref(n)))
    plain = out[: n.value].tobytes()
    m = O.OracleMphf(pf13_path())
    assert np.array_equal(O.count13(m, plain, 0), O.count13(m, buf, -1))


def test_normalize_kmer_counter_rules(gold):
    buf = open(os.path.join(gold, "kmer_counter", "mixed.fa"), "rb").read()
    a = np.frombuffer(buf, dtype=np.uint8)
    out = np.empty(a.shape[0] + 2, dtype=np.uint8)
    n = C.c_uint64()
    _lib.check(_lib.lib().aix_normalize_reads(a.ctypes.data_as(C.c_void_p), a.shape[0], 1, 1, out.ctypes.data_as(C.c_void_p), C.byref(n)))
    lines = out[: n.value].tobytes().split(b"\n")
    assert lines[-1] == b"" and all(b">" not in ln and b"\r" not in ln for ln in lines)
    assert len(lines) - 1 == buf.count(b">")


def test_header_is_plain_c_and_links(tmp_path):
    """include/aindex_hip.h compiles as C11 and every declared function resolves against libaindex_hip.so."""
    import subprocess
    src = tmp_path / "abi_check.c"
    calls = "\n".join(f"    p[{i}] = (void*){name};" for i, name in enumerate(_lib.header_symbols()))
    src.write_text('#include "aindex_hip.h"\n#include <stdio.h>\nint main(void) {\n    void* p[%d];\n%s\n'
                   '    int n = 0; int st = aix_device_count(&n);\n    printf("%%s %%d %%d\\n", aix_version(), st, (int)(sizeof(p) / sizeof(p[0])));\n    return 0;\n}\n'
                   % (len(_lib.header_symbols()), calls))
    exe = tmp_path / "abi_check"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-Wno-pedantic", f"-I{os.path.join(_lib.ROOT, 'include')}", str(src), "-o", str(exe),
                           f"-L{libdir}", "-laindex_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, timeout=120).stdout.decode()
    assert "gfx950" in out


# ------------------------------------------------------------------------------------------------
# host-side glue of the Python mirrors (no GPU needed)
# ------------------------------------------------------------------------------------------------
def test_list_packing_helper_matches_python_rules():
    """csrc/aix_pyfast.c (built next to the library) and the pure-Python path agree on what is a fixed-length batch."""
    from aindex_amd import wrapper
    f = wrapper.AindexWrapper._join_fixed
    cases = [(["ACG", "TTT"], 3, b"ACGTTT"), ([b"ACG", "TTT"], 3, b"ACGTTT"), (("ACG",), 3, b"ACG"), (["AC", "TTTT"], 3, None),
             (["ACG", 5], 3, None), (["AC\xe9"], 3, b"AC\xe9"), (["ACG", "TT€"], 3, None), (["ACGTACGTACGTA"] * 1000, 13, b"ACGTACGTACGTA" * 1000)]
    for helper in (True, False):
        saved = wrapper._PYFAST
        if not helper:
            wrapper._PYFAST = None                       # force the Python path
        try:
            if helper and wrapper._pyfast() is None:
                pytest.skip("aix_pyfast not built")
            for items, k, want in cases:
                got = f(items, k)
                # a list mixing str and bytes is packed by the helper and sent down the exact per-item path by the Python rules
                assert got == want or (not helper and got is None and len({type(x) for x in items}) > 1), (helper, items)
        finally:
            wrapper._PYFAST = saved


def test_list_packing_into_a_callers_buffer():
    """join_fixed_into (the wrapper's pinned staging is such a buffer): same bytes as join_fixed, None for an odd item, an error for a short buffer."""
    from aindex_amd import wrapper
    fast = wrapper._pyfast()
    if fast is None:
        pytest.skip("aix_pyfast not built")
    rng = np.random.default_rng(9)
    items = ["".join(rng.choice(list("ACGTN"), size=23)) for _ in range(1000)] * 300     # 300 000 items: above the helper's thread threshold
    items[7] = items[7].encode()
    buf = np.zeros(len(items) * 23 + 5, dtype=np.uint8)
    assert fast.join_fixed_into(items, 23, buf, 8) is True
    assert buf[: len(items) * 23].tobytes() == fast.join_fixed(items, 23, 8) and not buf[len(items) * 23:].any()
    assert fast.join_fixed_into(tuple(items[:10]), 23, buf) is True and fast.join_fixed_into([], 23, buf) is True
    bad = list(items)
    bad[250_000] = "ACG"
    assert fast.join_fixed_into(bad, 23, buf, 8) is None and fast.join_fixed_into(["ACGTACGTACGTACGTACGTAC€"], 23, buf) is None
    with pytest.raises(ValueError):
        fast.join_fixed_into(items, 23, np.zeros(100, dtype=np.uint8))
    with pytest.raises((TypeError, BufferError, ValueError)):
        fast.join_fixed_into(items[:4], 23, bytes(200))                                 # read-only buffer


def test_read_interval_bisection_equals_linear_rule():
    """get_rid / get_start: the bisection used for sorted, disjoint .ridx intervals returns what the reference's linear scan
    (first interval with start <= pos + 1 and end + 1 >= pos, python_wrapper.cpp:66-74) returns, gaps and touching reads included."""
    from aindex_amd.wrapper import AindexWrapper
    for seed in range(4):
        rng = np.random.default_rng(seed)
        lens, gaps = rng.integers(1, 50, size=200), rng.integers(0, 3, size=200)
        st = np.cumsum(np.concatenate([[0], (lens + gaps)[:-1]])).astype(np.uint64)
        en = st + lens.astype(np.uint64) - np.uint64(1)
        w = AindexWrapper.__new__(AindexWrapper)
        w.aindex_loaded, w.n_reads = True, 200
        w._ridx_rid, w._ridx_start, w._ridx_end = np.arange(200, dtype=np.uint64), st, en
        for sorted_flag in (True, False):
            w._ridx_sorted = sorted_flag
            for pos in range(0, int(en[-1]) + 5):
                hit = np.nonzero((st <= np.uint64(pos + 1)) & (en + np.uint64(1) >= np.uint64(pos)))[0]
                assert w._interval(pos) == (int(hit[0]) if hit.shape[0] else None), (seed, sorted_flag, pos)


def test_small_python_helpers():
    from aindex_amd.aindex import AIndex, get_revcomp, hamming_distance
    assert hamming_distance("ACGTN", "ACCTA") == 1 and hamming_distance("AAAA", "AAAT") == 1 and hamming_distance("", "A") == 0
    assert get_revcomp("ACGTNacgtn~[]") == "[]~nacgtNACGT"
    assert AIndex._index_to_13mer(None, 0) == "A" * 13 and AIndex._index_to_13mer(None, 4 ** 13 - 1) == "T" * 13
    assert AIndex._index_to_13mer(None, 27) == "AAAAAAAAAACGT"


def test_file_write_roundtrip(tmp_path):
    """aix_file_write (host only): the mapped, multi-threaded writer behind the tools' binary images — sizes around its slicing thresholds,
    empty files, an unwritable path."""
    import numpy as np
    L = _lib.lib()
    rng = np.random.default_rng(3)
    for n in (0, 1, 4095, 4096, (8 << 20) - 1, (8 << 20) + 5, 40_000_003):
        a = rng.integers(0, 256, size=n, dtype=np.uint8)
        p = str(tmp_path / f"f{n}.bin")
        assert L.aix_file_write(p.encode(), a.ctypes.data_as(_lib.vp) if n else None, n) == 0
        assert np.array_equal(np.fromfile(p, dtype=np.uint8), a)
    assert L.aix_file_write(str(tmp_path / "no" / "such" / "dir.bin").encode(), None, 0) == -2
