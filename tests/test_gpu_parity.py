"""GPU parity: the HIP path (through the C ABI) against the oracle / golden fixtures. Bit-exact.

Run on the MI355X box with `pytest -m gpu`. Nothing here reads /root/reference.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_lib as O
from aindex_amd import _lib, builder, synth
from aindex_amd.engine import Index


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(gold, *p):
    return json.load(open(os.path.join(gold, *p)))


def flat(strs):
    return np.frombuffer("".join(strs).encode(), dtype=np.uint8)


# ------------------------------------------------------------------------------------------------
# golden small23 index (built by the reference pipeline; stored set is NOT all-canonical)
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ix23(small23_prefix):
    ix = Index.open_23(small23_prefix + ".pf", small23_prefix + ".tf.bin", small23_prefix + ".kmers.bin")
    yield ix
    ix.close()


@pytest.fixture(scope="module")
def q23(gold):
    return load(gold, "small23", "queries.json")


def test_info(ix23, q23):
    i = ix23.info
    assert i["k"] == 23 and i["n"] == q23["n_kmers"] and i["bitpairs"] == 3 * i["hash_domain"]
    assert i["canonical_only"] == 0        # kmer_counter's pseudo-canonical keys


def test_q23_golden(ix23, q23):
    qs = flat(q23["queries"])
    assert ix23.tf_ascii(qs).tolist() == q23["tf"]
    assert ix23.total_ascii(qs).tolist() == q23["total"]
    f, r = ix23.both_ascii(qs)
    assert [[int(a), int(b)] for a, b in zip(f, r)] == q23["both"]
    assert ix23.hash_ascii(qs).tolist() == q23["hash"]
    kid, strand = ix23.kid_strand_ascii(qs)
    assert kid.tolist() == q23["kid"]
    assert strand.tolist() == q23["strand"]


def test_q23_ragged_matches_oracle(ix23, q23, small23_prefix):
    orc = O.OracleIndex23.from_prefix(small23_prefix)
    base = q23["queries"][:40] + q23["queries"][300:340]
    items = []
    for i, s in enumerate(base):
        items += [s, s + "ACGT"[: i % 5], s[: 23 - (i % 4)], s + s, ""]
    want = [orc.tf(x.encode()) for x in items]
    assert ix23.tf_ragged(items).tolist() == want
    assert any(want)


def test_q23_coverage_golden(ix23, q23):
    for cutoff in (0, 3):
        cases = [c for c in q23["coverage"] if c["cutoff"] == cutoff]
        got = ix23.coverage([c["seq"] for c in cases], cutoff)
        for c, g in zip(cases, got):
            assert g.tolist() == c["cov"]


def test_tf_and_checker_roundtrip(ix23, small23_prefix):
    assert np.array_equal(ix23.tf_array(), np.fromfile(small23_prefix + ".tf.bin", dtype=np.uint32))
    assert np.array_equal(ix23.checker_array(), np.fromfile(small23_prefix + ".kmers.bin", dtype=np.uint64))


def test_index_scatter_equals_compute_index(small23_prefix):
    rows = [ln.split("\t") for ln in open(small23_prefix + ".dat").read().split("\n") if ln]
    keys = flat([r[0] for r in rows])
    tfs = np.array([int(r[1]) for r in rows], dtype=np.uint32)
    pf = np.frombuffer(open(small23_prefix + ".pf", "rb").read(), dtype=np.uint8)
    n = len(rows)
    checker = np.empty(n, dtype=np.uint64)
    tf = np.empty(n, dtype=np.uint32)
    vp = _lib.vp
    _lib.check(_lib.lib().aix_index_scatter(pf.ctypes.data_as(vp), pf.shape[0], keys.ctypes.data_as(vp), tfs.ctypes.data_as(vp), n, 0,
                                            checker.ctypes.data_as(vp), tf.ctypes.data_as(vp)))
    assert np.array_equal(checker, np.fromfile(small23_prefix + ".kmers.bin", dtype=np.uint64))
    assert np.array_equal(tf, np.fromfile(small23_prefix + ".tf.bin", dtype=np.uint32))
    # a key outside the MPHF set collides with a stored one
    bad = keys.copy()
    bad[:23] = np.frombuffer(b"ACGTTGCATTTTACGATCGGCAT", dtype=np.uint8)
    st = _lib.lib().aix_index_scatter(pf.ctypes.data_as(vp), pf.shape[0], bad.ctypes.data_as(vp), tfs.ctypes.data_as(vp), n, 0,
                                      checker.ctypes.data_as(vp), tf.ctypes.data_as(vp))
    assert st in (0, -12)   # lands on a free slot only if it happens to hash onto its victim's slot


def test_index_scatter_shards_merge_to_whole(small23_prefix):
    """aix_index_scatter_shard: three uneven shards of the key set into full-size arrays; sum / max / OR == one scatter
    == the reference's files; a shard with an internal duplicate reports AIX_ERR_CONFLICT and still returns its arrays."""
    from aindex_amd._lib import lib, vp, AIX_ERR_CONFLICT
    pf = np.fromfile(small23_prefix + ".pf", dtype=np.uint8)
    checker = np.fromfile(small23_prefix + ".kmers.bin", dtype=np.uint64)
    tf = np.fromfile(small23_prefix + ".tf.bin", dtype=np.uint32)
    n = checker.shape[0]
    perm = np.random.default_rng(9).permutation(n)
    keys, cnts = synth.decode_kmers(checker, 23)[perm], tf[perm]
    acc_c, acc_t, acc_o = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n + 31) // 32, np.uint32)
    for lo, hi in ((0, 1), (1, n // 3), (n // 3, n), (n, n)):
        k = np.ascontiguousarray(keys[lo:hi]); c = np.ascontiguousarray(cnts[lo:hi])
        oc, ot, oo = np.empty(n, np.uint64), np.empty(n, np.uint32), np.empty((n + 31) // 32, np.uint32)
        st = lib().aix_index_scatter_shard(pf.ctypes.data_as(vp), pf.shape[0], k.ctypes.data_as(vp), c.ctypes.data_as(vp), hi - lo, n, 0,
                                           oc.ctypes.data_as(vp), ot.ctypes.data_as(vp), oo.ctypes.data_as(vp))
        assert st == 0
        assert int(np.unpackbits(oo.view(np.uint8)).sum()) == hi - lo and not np.any(acc_o & oo)
        acc_c = np.maximum(acc_c, oc); acc_t += ot; acc_o |= oo
    assert np.array_equal(acc_c, checker) and np.array_equal(acc_t, tf)
    assert int(np.unpackbits(acc_o.view(np.uint8)).sum()) == n
    k = np.ascontiguousarray(np.concatenate([keys[:5], keys[:1]]))
    oc, ot, oo = np.empty(n, np.uint64), np.empty(n, np.uint32), np.empty((n + 31) // 32, np.uint32)
    st = lib().aix_index_scatter_shard(pf.ctypes.data_as(vp), pf.shape[0], k.ctypes.data_as(vp), None, 6, n, 0,
                                       oc.ctypes.data_as(vp), ot.ctypes.data_as(vp), oo.ctypes.data_as(vp))
    assert st == AIX_ERR_CONFLICT and int(np.unpackbits(oo.view(np.uint8)).sum()) == 5 and not ot.any()      # mock mode: tf stays 0


def test_count23_fixed_refx86_equals_pipeline(ix23, gold, small23_prefix):
    fa = open(os.path.join(gold, "small23", "reads.fa"), "rb").read()
    tf = ix23.count23_fixed(fa, _lib.FMT_FASTA, _lib.CANON_REF_X86)
    assert np.array_equal(tf, np.fromfile(small23_prefix + ".tf.bin", dtype=np.uint32))
    orc = O.OracleIndex23.from_prefix(small23_prefix)
    for mode in (0, 2):
        assert np.array_equal(ix23.count23_fixed(fa, _lib.FMT_FASTA, mode), orc.count23_fixed(fa, True, mode))


# ------------------------------------------------------------------------------------------------
# synthetic true-canonical index (canonical fast path), mid size, vs oracle
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def canon_case(tmp_path_factory):
    d = tmp_path_factory.mktemp("canon")
    g = synth.genome_codes(23, 300_000)
    keys, counts = synth.canonical_distinct(g, 23)
    pf = builder.build_pf_codes(keys, 23)
    prefix = str(d / "canon")
    open(prefix + ".pf", "wb").write(pf)
    m = O.OracleMphf(prefix + ".pf")
    ascii_keys = synth.decode_kmers(keys, 23)
    # product-side scatter makes the files; the oracle then loads them like the reference would
    n = keys.shape[0]
    checker = np.empty(n, dtype=np.uint64)
    tf = np.empty(n, dtype=np.uint32)
    vp = _lib.vp
    pfa = np.frombuffer(pf, dtype=np.uint8)
    flatk = np.ascontiguousarray(ascii_keys).reshape(-1)
    _lib.check(_lib.lib().aix_index_scatter(pfa.ctypes.data_as(vp), pfa.shape[0], flatk.ctypes.data_as(vp), counts.ctypes.data_as(vp), n, 0,
                                            checker.ctypes.data_as(vp), tf.ctypes.data_as(vp)))
    rc, ochecker, otf = O.index_scatter(m, flatk, counts)
    assert rc == 0 and np.array_equal(checker, ochecker) and np.array_equal(tf, otf)
    checker.tofile(prefix + ".kmers.bin")
    tf.tofile(prefix + ".tf.bin")
    ix = Index.open_23(prefix + ".pf", prefix + ".tf.bin", prefix + ".kmers.bin")
    orc = O.OracleIndex23.from_prefix(prefix)
    yield {"ix": ix, "orc": orc, "genome": g, "keys": keys, "counts": counts, "prefix": prefix}
    ix.close()


def mixed_queries(genome_codes, n, seed):
    """50 % genome windows on a random strand, 50 % uniform random 23-mers, a sprinkle of N / lower-case."""
    w = synth.rolling_codes(genome_codes, 23)
    pick = (synth.sm64(seed, np.arange(n // 2, dtype=np.uint64)) % np.uint64(w.shape[0])).astype(np.int64)
    codes = w[pick]
    flip = (synth.sm64(seed + 1, np.arange(n // 2, dtype=np.uint64)) & np.uint64(1)).astype(bool)
    codes = np.where(flip, synth.revcomp_codes(codes, 23), codes)
    q = np.concatenate([synth.decode_kmers(codes, 23), synth.random_kmers_ascii(seed + 2, n - n // 2, 23)])
    q = q.copy()
    idx = np.arange(0, q.shape[0], 97)
    q[idx, (idx * 7) % 23] = ord("N")
    idx = np.arange(5, q.shape[0], 131)
    q[idx] |= 0x20                      # lower-case letters
    return q


def test_canonical_index_flag_and_queries(canon_case):
    ix, orc = canon_case["ix"], canon_case["orc"]
    assert ix.canonical_only
    q = mixed_queries(canon_case["genome"], 200_000, 8)
    want = orc.tf_batch(q, threads=8)
    assert (want != 0).sum() > 50_000
    got_fast = ix.tf_ascii(q)
    assert np.array_equal(got_fast, want)
    ix.set_canonical_fastpath(False)
    try:
        assert np.array_equal(ix.tf_ascii(q), want)
        tot_slow = ix.total_ascii(q)
        kid_slow, strand_slow = ix.kid_strand_ascii(q)
    finally:
        ix.set_canonical_fastpath(True)
    assert np.array_equal(ix.total_ascii(q), tot_slow)
    kid, strand = ix.kid_strand_ascii(q)
    assert np.array_equal(kid, kid_slow) and np.array_equal(strand, strand_slow)
    sub = q[:3000]
    assert tot_slow[:3000].tolist() == [orc.total(bytes(s)) for s in sub]
    assert strand[:3000].tolist() == [orc.strand(bytes(s)) for s in sub]
    assert kid[:3000].tolist() == [orc.kid(bytes(s)) for s in sub]
    assert np.array_equal(ix.hash_ascii(q), orc.hash_batch(q))


def test_codes_api_equals_ascii(canon_case):
    ix = canon_case["ix"]
    q = synth.decode_kmers(synth.rolling_codes(canon_case["genome"][:50_000], 23), 23)
    codes = synth.encode_kmers(q)
    a = ix.tf_ascii(q)
    assert np.array_equal(ix.tf_codes(codes), a)
    ix.set_canonical_fastpath(False)
    try:
        assert np.array_equal(ix.tf_codes(codes), a)
    finally:
        ix.set_canonical_fastpath(True)
    assert (a != 0).all()


def test_coverage_vs_oracle(canon_case):
    ix, orc, g = canon_case["ix"], canon_case["orc"], canon_case["genome"]
    asc = synth.genome_ascii(23, 300_000)
    seqs = [bytes(asc[1000:6000]), bytes(asc[20000:20030]), b"ACGT", b"", bytes(asc[7000:7023]), bytes(asc[9000:12000]).lower(),
            bytes(asc[40000:41000]).replace(b"A", b"N", 3)]
    for cutoff in (0, 2):
        got = ix.coverage(seqs, cutoff)
        for s, gq in zip(seqs, got):
            assert np.array_equal(gq, orc.coverage(s, cutoff))


def test_count23_true_rc_vs_oracle_and_genome_multiplicity(canon_case):
    ix, orc = canon_case["ix"], canon_case["orc"]
    asc = synth.genome_ascii(23, 300_000)
    reads = synth.reads_plain(41, asc, 3000, 150, rc_fraction_half=True, n_rate_ppm=1000).tobytes()
    got = ix.count23_fixed(reads, _lib.FMT_PLAIN, _lib.CANON_TRUE_RC)
    assert np.array_equal(got, orc.count23_fixed(reads, False, 2))
    assert got.sum() > 300_000
    # the whole genome as one read reproduces the stored multiplicities (tf = genome multiplicity)
    whole = ix.count23_fixed(bytes(asc) + b"\n", _lib.FMT_PLAIN, _lib.CANON_TRUE_RC)
    assert np.array_equal(whole, ix.tf_array())


# ------------------------------------------------------------------------------------------------
# 13-mer mode
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ix13():
    from pf13 import pf13_path
    ix = Index.open_13(pf13_path(), None)
    yield ix
    ix.close()


NAMES13 = ["refdata_test.fasta", "refdata_test_se.fastq", "refdata_test_reads.txt", "refdata_test_unknown.txt",
           "refdata_test_R1.fastq", "synth.fa", "synth.fq", "synth.txt"]


@pytest.mark.parametrize("name", NAMES13)
def test_count13_golden(ix13, gold, name):
    z = np.load(os.path.join(gold, "count13", "expected.npz"))
    buf = open(os.path.join(gold, "count13", name), "rb").read()
    counts = ix13.count13(buf)
    nz = np.nonzero(counts)[0]
    assert np.array_equal(nz.astype(np.uint64), z[name + ".idx"])
    assert np.array_equal(counts[nz], z[name + ".cnt"])


def test_q13_golden(ix13, gold):
    z = np.load(os.path.join(gold, "count13", "expected.npz"))
    tf = np.zeros(4 ** 13, dtype=np.uint64)
    tf[z["synth.fa.idx"].astype(np.int64)] = z["synth.fa.cnt"]
    tf[int(z["synth.fa.idx"][0])] += 1 << 32          # exercises the u64 -> u32 truncation of get_tf_values
    ix13.set_tf_13(tf)
    q = load(gold, "q13.json")
    want = list(q["tf"])
    got = ix13.tf_ragged(q["queries"]).tolist()
    assert got == want
    valid = flat(q["valid"])
    tot = ix13.total_ascii(valid)
    f, r = ix13.both_ascii(valid)
    hot = int(tf[int(z["synth.fa.idx"][0])])
    exp_tot = np.array(q["total"], dtype=np.uint64)
    exp_both = np.array(q["both"], dtype=np.uint64)
    # undo the 2^32 bump for the comparison with the golden (which was taken without it)
    bump_f = (f >= (1 << 32))
    bump_r = (r >= (1 << 32))
    assert np.array_equal(f - bump_f.astype(np.uint64) * np.uint64(1 << 32), exp_both[:, 0])
    assert np.array_equal(r - bump_r.astype(np.uint64) * np.uint64(1 << 32), exp_both[:, 1])
    assert np.array_equal(tot, f + r)
    assert np.array_equal(ix13.tf_array(), tf)
    for c in q["coverage"]:
        assert ix13.coverage([c["seq"]], c["cutoff"])[0].tolist() == c["cov"]
    assert hot >= (1 << 32)


def test_13mer_vs_oracle_synthetic(ix13):
    from pf13 import pf13_path
    asc = synth.genome_ascii(13, 400_000)
    reads = synth.reads_plain(14, asc, 20_000, 150, n_rate_ppm=1000).tobytes()
    m = O.OracleMphf(pf13_path())
    want = O.count13(m, reads, 0, threads=8)
    got = ix13.count13(reads, _lib.FMT_PLAIN)
    assert np.array_equal(got, want)
    assert int(got.sum()) > 2_500_000
    ix13.set_tf_13(got)
    orc = O.OracleIndex13(pf13_path(), got)
    q = np.concatenate([synth.decode_kmers(synth.rolling_codes(synth.genome_codes(13, 400_000)[:100_000], 13), 13),
                        synth.random_kmers_ascii(15, 100_000, 13)]).copy()
    q[::50, 3] = ord("N")
    q[7::61] |= 0x20
    assert np.array_equal(ix13.tf_ascii(q), orc.tf_batch(q, threads=8))
    sub = q[:4000]
    tot = ix13.total_ascii(sub)
    assert tot.tolist() == [orc.total(bytes(s)) for s in sub]
    f, r = ix13.both_ascii(sub)
    assert [(int(a), int(b)) for a, b in zip(f, r)] == [orc.both(bytes(s)) for s in sub]


# ------------------------------------------------------------------------------------------------
# device generators == numpy mirror; HBM-resident entry points == host entry points
# ------------------------------------------------------------------------------------------------
def test_synth_generators_match_numpy():
    import torch
    from aindex_amd import engine
    g = engine.synth_genome_t(23, 100_003)
    assert np.array_equal(g.cpu().numpy(), synth.genome_ascii(23, 100_003))
    for k in (23, 13):
        q = engine.synth_kmers_t(7, 5000, k, first=11)
        assert np.array_equal(q.cpu().numpy().reshape(-1, k), synth.random_kmers_ascii(7, 5000, k, start=11))
    r = engine.synth_reads_t(41, g, 2000, 150, rc_half=True, n_rate_ppm=1000, first_read=5)
    want = synth.reads_plain(41, synth.genome_ascii(23, 100_003), 2000, 150, rc_fraction_half=True, n_rate_ppm=1000, first_read=5)
    assert np.array_equal(r.cpu().numpy(), want)
    torch.cuda.synchronize()


def test_device_entry_points(canon_case):
    import torch
    from aindex_amd import engine
    ix = canon_case["ix"]
    q = engine.synth_kmers_t(7, 100_000, 23)
    out = ix.tf_ascii_t(q)
    torch.cuda.synchronize()
    host = ix.tf_ascii(q.cpu().numpy())
    assert np.array_equal(out.cpu().numpy().view(np.uint32), host)
    # unaligned device views must work too (window loader never reads outside the view)
    buf = torch.empty(23 * 1000 + 64, dtype=torch.uint8, device="cuda")
    for off in (1, 2, 3):
        v = buf[off: off + 23 * 1000]
        v.copy_(q[: 23 * 1000])
        o = ix.tf_ascii_t(v)
        torch.cuda.synchronize()
        assert np.array_equal(o.cpu().numpy().view(np.uint32), host[:1000])


# ------------------------------------------------------------------------------------------------
# size-independent properties at a larger size (no oracle in the loop)
# ------------------------------------------------------------------------------------------------
def test_properties_at_scale(canon_case):
    import torch
    from aindex_amd import engine
    ix = canon_case["ix"]
    n = 5_000_000
    q = engine.synth_kmers_t(70, n, 23)
    a = ix.tf_ascii_t(q).cpu().numpy().view(np.uint32)
    ix.set_canonical_fastpath(False)
    try:
        b = ix.tf_ascii_t(q).cpu().numpy().view(np.uint32)
    finally:
        ix.set_canonical_fastpath(True)
    assert np.array_equal(a, b)                       # one-probe fast path == reference two-probe order
    assert (a != 0).sum() == 0 or (a != 0).mean() < 1e-3   # uniform random 23-mers essentially never hit
    # every stored key and its reverse complement return the stored tf
    keys, counts = canon_case["keys"], canon_case["counts"]
    tf_fwd = ix.tf_codes(keys)
    tf_rc = ix.tf_codes(synth.revcomp_codes(keys, 23))
    assert np.array_equal(tf_fwd, counts) and np.array_equal(tf_rc, counts)
    tot = ix.total_ascii(synth.decode_kmers(keys[:100_000], 23))
    assert np.array_equal(tot, 2 * counts[:100_000].astype(np.uint64))


# ------------------------------------------------------------------------------------------------
# A1/A2 positions index and the Python mirrors
# ------------------------------------------------------------------------------------------------
def test_positions_fill_equals_compute_aindex(ix23, gold, small23_prefix):
    z = np.load(os.path.join(gold, "small23", "aindex.npz"))
    reads = open(small23_prefix + ".reads", "rb").read()
    indices, pos = ix23.positions_fill(reads)
    assert np.array_equal(indices, z["indices"])
    assert np.array_equal(pos, z["index"])            # the reference's 1-thread run, slot for slot


def test_positions_fill_device_twin(ix23, gold, small23_prefix):
    """aix_positions_fill_dev (reads, indices, positions in HBM) == the host-buffer call == the reference's files; the
    start adjustment is computed from the head the caller fetches."""
    import torch
    z = np.load(os.path.join(gold, "small23", "aindex.npz"))
    reads = open(small23_prefix + ".reads", "rb").read()
    for buf, want in ((reads, (z["indices"], z["index"])), (b"?AC\n" + reads[:9000], None), (b"ACGT\n" * 20000 + reads[:5000], None), (b"", None)):
        if want is None:
            want = O.OracleIndex23.from_prefix(small23_prefix).positions(buf)
        t = torch.frombuffer(bytearray(buf) if buf else bytearray(1), dtype=torch.uint8)[: len(buf)].cuda()
        ind, pos = ix23.positions_fill_t(t)
        assert np.array_equal(ind.cpu().numpy().view(np.uint64), want[0]) and np.array_equal(pos.cpu().numpy().view(np.uint64), want[1])


def test_positions_fill_vs_oracle_true_canonical(canon_case):
    ix, orc = canon_case["ix"], canon_case["orc"]
    asc = synth.genome_ascii(23, 300_000)
    r = synth.reads_plain(41, asc, 4000, 150, rc_fraction_half=True, n_rate_ppm=2000).reshape(-1, 151).copy()
    r[::7, 150] = ord("~")                            # PE separator inside a line
    r[5::50, 10] = ord("?")
    r[3::40] |= 0x20                                   # lower-case reads: raw bytes hashed on the forward strand
    r[3::40, 150] = ord("\n")
    reads = b"?AC\n" + r.tobytes()                    # exercises the start-adjust quirk (hash.cpp:973-986)
    want_ind, want_pos = orc.positions(reads)
    indices, pos = ix.positions_fill(reads)
    assert np.array_equal(indices, want_ind)
    assert np.array_equal(pos, want_pos)
    assert (pos != 0).sum() > 100_000


@pytest.mark.parametrize("piece", [1, 7, 1000, 50_000])
def test_positions_fill_in_pieces(ix23, gold, small23_prefix, canon_case, piece):
    """Buffers of more than 2^30 windows are filled piece by piece with the per-bucket fill counters carried over;
    AIX_POSITIONS_PIECE shrinks the piece so that this runs at test sizes. Every piece size must reproduce the reference's
    1-thread files (golden) and the oracle (tf-capped buckets, start quirk, separators on piece borders)."""
    os.environ["AIX_POSITIONS_PIECE"] = str(piece)
    try:
        z = np.load(os.path.join(gold, "small23", "aindex.npz"))
        reads = open(small23_prefix + ".reads", "rb").read()
        if piece < 1000:
            reads = reads[: 6000]
            want_ind, want_pos = O.OracleIndex23.from_prefix(small23_prefix).positions(reads)
        else:
            want_ind, want_pos = z["indices"], z["index"]
        indices, pos = ix23.positions_fill(reads)
        assert np.array_equal(indices, want_ind) and np.array_equal(pos, want_pos)
        if piece >= 1000:
            ix, orc = canon_case["ix"], canon_case["orc"]
            asc = synth.genome_ascii(23, 300_000)
            r = synth.reads_plain(43, asc, 1500, 150, rc_fraction_half=True, n_rate_ppm=2000).reshape(-1, 151).copy()
            r[::7, 150] = ord("~")
            r[3::40] |= 0x20
            r[3::40, 150] = ord("\n")
            buf = b"?AC\n" + r.tobytes() + r[:700].tobytes()            # repeated reads overflow their buckets (slot >= tf is dropped)
            want_ind, want_pos = orc.positions(buf)
            indices, pos = ix.positions_fill(buf)
            assert np.array_equal(indices, want_ind) and np.array_equal(pos, want_pos)
    finally:
        del os.environ["AIX_POSITIONS_PIECE"]


@pytest.mark.parametrize("cap", [0, 40, 3])
def test_positions_fill_msd_path_equals_sort_path_and_oracle(ix23, gold, small23_prefix, canon_case, tmp_path, cap):
    """A2 without the library sort (aix_a2msd.hip: two-level MSD partition of the (bucket, offset) pairs + per-bucket LDS stage)
    is the default from 2^22 windows up; AIX_A2_MSD=1 forces it at test sizes, AIX_A2_TEST_CAP shrinks what a bucket may hold in
    LDS so that buckets are set aside and take the gather + radix-sort path (cap 3: nearly all of them). Every variant must give
    the reference's 1-thread files (golden), the oracle (tf-capped buckets, repeats, start quirk, separators, lower case), also
    piece by piece (fill counters carried over), and a slot with hundreds of pairs (cooperative in-LDS sort) with tf above and
    below its occurrences."""
    env = {"AIX_A2_MSD": "1"}
    if cap:
        env["AIX_A2_TEST_CAP"] = str(cap)
    os.environ.update(env)
    try:
        z = np.load(os.path.join(gold, "small23", "aindex.npz"))
        reads = open(small23_prefix + ".reads", "rb").read()
        indices, pos = ix23.positions_fill(reads)
        assert np.array_equal(indices, z["indices"]) and np.array_equal(pos, z["index"])
        ix, orc = canon_case["ix"], canon_case["orc"]
        asc = synth.genome_ascii(23, 300_000)
        r = synth.reads_plain(47, asc, 3000, 150, rc_fraction_half=True, n_rate_ppm=2000).reshape(-1, 151).copy()
        r[::7, 150] = ord("~")
        r[5::50, 10] = ord("?")
        r[3::40] |= 0x20
        r[3::40, 150] = ord("\n")
        buf = b"?AC\n" + r.tobytes() + r[:900].tobytes()                  # repeated reads overflow their buckets (slot >= tf is dropped)
        want_ind, want_pos = orc.positions(buf)
        indices, pos = ix.positions_fill(buf)
        assert np.array_equal(indices, want_ind) and np.array_equal(pos, want_pos) and (pos != 0).sum() > 100_000
        assert ix.info["positions_backend"] == 2                             # which grouping ran is reported, never silent
        os.environ["AIX_POSITIONS_PIECE"] = "70000"
        try:
            indices, pos = ix.positions_fill(buf)
        finally:
            del os.environ["AIX_POSITIONS_PIECE"]
        assert np.array_equal(pos, want_pos)
        os.environ["AIX_A2_TEST_NOMEM"] = "1"                              # the MSD workspace "does not fit": the piece takes the sort path, same answer
        try:
            assert np.array_equal(ix.positions_fill(buf)[1], want_pos)
            assert ix.info["positions_backend"] == 1                         # the sort path, and the handle says so
        finally:
            del os.environ["AIX_A2_TEST_NOMEM"]
        # the shard protocol on top of it (fill counters of earlier shards handed in, file-relative offsets): three shards == the whole
        from shard_helpers import positions_by_shards
        sind, acc, _first = positions_by_shards(ix, buf, 3)
        assert np.array_equal(sind, want_ind) and np.array_equal(acc, want_pos)
        # heavy slots: one read 700 times (its k-mers: 700 pairs each), tf raised to 1000 for half of them and to 150 for the rest
        prefix = canon_case["prefix"]
        checker = np.fromfile(prefix + ".kmers.bin", dtype=np.uint64)
        tf = np.fromfile(prefix + ".tf.bin", dtype=np.uint32)
        one = r[11:12].copy()
        one[0, :150] = np.frombuffer(asc[1000:1150].tobytes(), dtype=np.uint8)
        one[0, 150] = ord("\n")
        codes = synth.encode_kmers(np.lib.stride_tricks.sliding_window_view(one[0, :150], 23))
        canon = np.minimum(codes, synth.revcomp_codes(codes, 23))
        slots = np.flatnonzero(np.isin(checker, canon))
        assert slots.shape[0] == 128
        tf2 = tf.copy()
        tf2[slots[::2]] = 1000
        tf2[slots[1::2]] = 150
        p2 = str(tmp_path / "heavy")
        import shutil
        shutil.copy(prefix + ".pf", p2 + ".pf")
        checker.tofile(p2 + ".kmers.bin")
        tf2.tofile(p2 + ".tf.bin")
        orc2 = O.OracleIndex23.from_prefix(p2)
        heavy = r[:500].tobytes() + one.tobytes() * 700 + r[500:800].tobytes()
        want_ind, want_pos = orc2.positions(heavy)
        with Index.open_23(p2 + ".pf", p2 + ".tf.bin", p2 + ".kmers.bin") as ixh:
            indices, pos = ixh.positions_fill(heavy)
            assert np.array_equal(indices, want_ind) and np.array_equal(pos, want_pos)
            os.environ["AIX_A2_MSD"] = "0"                                 # and the sort path on the same input
            i0, p0 = ixh.positions_fill(heavy)
            assert np.array_equal(p0, want_pos)
    finally:
        for k in list(env) + ["AIX_A2_MSD"]:
            os.environ.pop(k, None)


@pytest.mark.parametrize("world", [2, 3, 5])
def test_positions_fill_shards_equal_whole(canon_case, world):
    """The two-pass shard protocol of dist.positions_fill_sharded, run rank by rank in one process: per-shard bucket
    tallies, exclusive sum over earlier shards, shard fills into full-size arrays, sum == the unsharded fill == oracle.
    The buffer starts with a stretch without any clean window (the start adjustment carries into later shards), holds
    '?' right after it, lower-case reads, PE separators and repeated reads that overflow their buckets."""
    from shard_helpers import positions_by_shards
    ix, orc = canon_case["ix"], canon_case["orc"]
    asc = synth.genome_ascii(23, 300_000)
    r = synth.reads_plain(45, asc, 600, 150, rc_fraction_half=True, n_rate_ppm=2000).reshape(-1, 151).copy()
    r[::7, 150] = ord("~")
    r[3::40] |= 0x20
    r[3::40, 150] = ord("\n")
    r[0, 5] = ord("?")
    body = r.tobytes() + r[:300].tobytes()
    for lead in (b"", b"ACGTAC\n" * (2 * len(body) // 7), b"?AC\n"):
        buf = lead + body
        want_ind, want_pos = orc.positions(buf)
        assert np.array_equal(ix.positions_fill(buf)[1], want_pos)
        sind, acc, first = positions_by_shards(ix, buf, world)
        assert np.array_equal(sind, want_ind)
        assert np.array_equal(ix.positions_indices(), want_ind) and np.array_equal(acc, want_pos), (world, len(lead))
        if lead.startswith(b"ACGTAC"):
            assert first[1] is True                                       # the adjustment did carry into the second shard


@pytest.mark.slow
def test_positions_fill_beyond_4gib(canon_case):
    """4.53 GB of reads (> 2^32 bytes): five pieces (default 2^30 windows) == three pieces (2^31); every stored offset
    points at a window whose numerically smaller strand is the bucket's key, ascending inside the bucket, and a bucket is
    short only when the buffer holds fewer than tf occurrences (none here)."""
    from aindex_amd import engine
    ix = canon_case["ix"]
    g = engine.synth_genome_t(23, 300_000)
    reads_t = engine.synth_reads_t(44, g, 30_000_000, 150, rc_half=True, n_rate_ppm=500)
    assert reads_t.numel() > (1 << 32)
    reads = reads_t.cpu().numpy()
    del reads_t
    buf = reads.tobytes()
    ind, pos = ix.positions_fill(buf)
    os.environ["AIX_POSITIONS_PIECE"] = str(1 << 31)
    try:
        ind2, pos2 = ix.positions_fill(buf)
    finally:
        del os.environ["AIX_POSITIONS_PIECE"]
    assert np.array_equal(ind, ind2) and np.array_equal(pos, pos2)
    tf = ix.tf_array()
    checker = ix.checker_array()
    assert int(ind[-1]) == int(tf.sum(dtype=np.uint64)) and np.all(pos != 0)            # 15 000x coverage fills every bucket
    rng = np.random.default_rng(5)
    for h in rng.integers(0, ix.n, size=300):
        p = pos[int(ind[h]): int(ind[h + 1])].astype(np.int64) - 1
        assert np.all(np.diff(p) > 0)
        w = np.stack([reads[q: q + 23] for q in p])
        codes = synth.encode_kmers(w)
        canon = np.minimum(codes, synth.revcomp_codes(codes, 23))
        assert np.all(canon == checker[h])
    # the first tf offsets of a bucket are the first occurrences in the buffer: nothing before pos[first] matches
    h = int(rng.integers(0, ix.n))
    first = int(pos[int(ind[h])]) - 1
    head = np.frombuffer(buf[: first + 22], dtype=np.uint8)
    if head.shape[0] >= 23 and first < 5_000_000:
        win = np.lib.stride_tricks.sliding_window_view(head, 23)
        ok = np.all((win == 65) | (win == 67) | (win == 71) | (win == 84), axis=1)
        c = synth.encode_kmers(win[ok])
        assert not np.any(np.minimum(c, synth.revcomp_codes(c, 23)) == checker[h])


def test_python_mirrors(gold, small23_prefix, tmp_path):
    from aindex_amd.aindex import AIndex, Strand
    q = load(gold, "small23", "queries.json")
    ai = AIndex.load_from_prefix(small23_prefix)                  # auto-detect picks 23 (.kmers.bin present)
    qs = q["queries"]
    assert ai.get_tf_values(qs) == q["tf"]
    assert [ai.get_tf_value(s) for s in qs[:20]] == q["tf"][:20]
    assert [ai[s] for s in qs[300:320]] == q["tf"][300:320]
    assert ai.get_hash_values(qs[:100]) == q["hash"][:100]
    assert [int(ai.get_strand(s)) for s in qs[290:310]] == q["strand"][290:310]
    assert [ai.get_kid_by_kmer(s) for s in qs[:30]] == q["kid"][:30]
    assert len(ai) == q["hash_size"] and ai.n_kmers == q["n_kmers"]
    w = ai._wrapper
    # the batch surface on every shape a batch can arrive in: list[str], list[bytes], a tuple, ONE buffer of N * 23 bytes
    # (bytes / bytearray / memoryview / numpy 'S23' / numpy uint8), ONE joined str; a batch above the thread threshold of the C helper; and
    # values above CPython's small-int cache (256) in the list the call returns
    fixed = [s for s in qs if len(s) == 23 and s.isascii()]
    want = [t for s, t in zip(qs, q["tf"]) if len(s) == 23 and s.isascii()]
    assert len(fixed) > 1000 and any(want)
    joined = "".join(fixed)
    for arg in (fixed, [s.encode() for s in fixed], tuple(fixed), joined.encode(), bytearray(joined.encode()), memoryview(joined.encode()),
                np.array([s.encode() for s in fixed], dtype="S23"), np.frombuffer(joined.encode(), dtype=np.uint8), joined):
        got = w.get_tf_values(arg)
        assert isinstance(got, list) and got == want, type(arg)
    assert w.get_tf_values(fixed[0]) == [want[0]] and w.get_tf_values(fixed[0][:20]) == [0] and w.get_tf_values([]) == [] and w.get_tf_values("") == []
    big = fixed * (250_000 // len(fixed) + 1)
    assert w.get_tf_values(big) == want * (250_000 // len(fixed) + 1)
    # lists of >= 4096 items go through the wrapper's pinned staging (kept between calls, grown when a larger batch arrives); a busy staging,
    # an odd item or a batch above AIX_STAGE_MAX_MB falls back to the unstaged path with the same answers
    reps = 250_000 // len(fixed) + 1
    assert w._stage._blk and w.get_tf_values(tuple(big)) == want * reps                 # second call: same blocks
    cap_before = w._stage._blk["in"][1]
    assert w.get_tf_values(big * 3) == want * (3 * reps) and w._stage._blk["in"][1] > cap_before
    with w._stage.lock:
        assert w.get_tf_values(big) == want * reps                                      # staging busy (another thread inside): unstaged path
    odd = list(big)
    odd[1234] = fixed[0][:20]                                                           # one 20-character item: the general (ragged) path answers 0 for it
    got = w.get_tf_values(odd)
    assert got[1234] == 0 and got[:1234] == (want * reps)[:1234] and got[1235:] == (want * reps)[1235:]
    from aindex_amd.wrapper import AindexWrapper as _W
    vals = np.array([0, 1, 255, 256, 257, 65536, 2 ** 32 - 1] * 40_000, dtype=np.uint32)
    lst = _W._to_list(vals)
    assert lst == vals.tolist() and all(type(v) is int for v in lst[:14])
    assert w.get_total_tf_values_23mer(qs) == q["total"]
    assert [list(p) for p in w.get_tf_both_directions_23mer_batch(qs)] == q["both"]
    for c in q["coverage"]:
        assert ai.get_sequence_coverage(c["seq"], c["cutoff"], 23) == c["cov"]
    kid = q["kid"][0]
    kmer = ai.get_kmer_by_kid(kid)
    assert kmer == qs[0] and ai.get_kmer_info(kid)[2] == q["tf"][0]
    assert Strand(1) == ai.get_strand(qs[0])
    # positions: build with the GPU, write the reference's files, query through the mirror
    prefix = str(tmp_path / "s23")
    indices, pos = w.build_aindex(small23_prefix + ".reads", prefix)
    z = np.load(os.path.join(gold, "small23", "aindex.npz"))
    assert np.array_equal(np.fromfile(prefix + ".index.bin", dtype=np.uint64), z["index"])
    assert np.array_equal(np.fromfile(prefix + ".indices.bin", dtype=np.uint64), z["indices"])
    w.load_aindex_from_prefix_23mer(prefix, 100)
    reads = open(small23_prefix + ".reads", "rb").read()
    found = 0
    for s in qs[:50] + qs[300:350]:
        got = w.get_positions(s)
        for p in got:
            win = reads[p:p + 23].decode()
            assert win == s or win == w.get_reverse_complement_23mer(s)
        found += len(got)
    assert found > 50
    assert w.get_positions("ACGT") == [] and w.get_positions(qs[700]) == []


def test_positions_and_reads_access_golden(gold, small23_prefix, tmp_path):
    """N2: get_positions / get_rid / get_start / get_read_by_rid / get_read through the AIndex mirror == the answers of the
    reference's compiled pybind module (tests/golden/small23/access.json, make_golden.py:make_small23_access), with the
    positions files built by the GPU; then the pure-Python helpers layered on them."""
    from aindex_amd.aindex import AIndex, hamming_distance
    a = json.load(open(os.path.join(gold, "small23", "access.json")))
    ai = AIndex.load_from_prefix(small23_prefix)
    prefix = str(tmp_path / "acc")
    ai._wrapper.build_aindex(small23_prefix + ".reads", prefix)
    ai.load_aindex(prefix + ".index.bin", prefix + ".indices.bin", 100)
    ai.load_reads(small23_prefix + ".reads")
    assert ai.n_reads == a["n_reads"] and ai.reads_size == a["reads_size"] and ai.get_reads_size() == a["reads_size"]
    assert [ai.get_positions(s) for s in a["kmers"]] == a["positions"]
    assert [ai.pos(s) for s in a["kmers"][:5]] == a["positions"][:5]
    assert [ai.get_rid(p) for p in a["probes"]] == a["rid"]
    assert [ai.get_start(p) for p in a["probes"]] == a["start"]
    for rid, read in a["read_by_rid"].items():
        assert ai.get_read_by_rid(int(rid)) == read
    for s0, e0, rcflag, want in a["get_read"]:
        assert ai.get_read(s0, e0, rcflag) == want, (s0, e0, rcflag)
    # helpers on top (aindex.py:271-343): every reported offset really holds the k-mer (or its reverse complement)
    i0 = max(range(len(a["kmers"])), key=lambda i: len(a["positions"][i]))
    k0 = a["kmers"][i0]
    hits = ai.get_rid2poses(k0)
    assert sum(len(v) for v in hits.values()) == len(a["positions"][i0]) > 0
    rc0 = ai._wrapper.get_reverse_complement_23mer(k0)
    for rid, offs in hits.items():
        read = ai.get_read_by_rid(rid)
        assert all(read[o:o + 23] in (k0, rc0) for o in offs)
    got = ai.get_reads_by_kmer(k0, 3)
    assert 0 < len(got) <= 3 and all(k0 in r or rc0 in r for r in got) and len(set(got)) == len(got)
    reads = list(ai.iter_reads())
    assert len(reads) == a["n_reads"] and reads[1] == (1, a["read_by_rid"]["1"])
    assert sum(1 for _ in ai.iter_reads_se()) == a["n_reads"]                  # single-end reads: one sub-read each
    assert ai.get_header(5) is None
    hdr = str(tmp_path / "h.header")
    open(hdr, "w").write("chr1.1 first\t0\t151\nchr2 second\t151\t151\n")
    ai.load_reads_index(small23_prefix + ".ridx", hdr)
    assert ai.get_header(150) == "chr1.1 first" and ai.get_header(151) == "chr2 second" and ai.get_header(10 ** 9) == ""
    assert ai.chrm2start == {"chr1": 0, "chr2": 151} and ai.rid2start[1] == (151, 301)
    assert hamming_distance("ACGTN", "ACCTA") == 1
    # k-mers by frequency: the reference's enumeration (kid order, get_tf_value of the stored k-mer), stable descending sort
    tf = np.fromfile(small23_prefix + ".tf.bin", dtype=np.uint32)
    kmers = [ai.get_kmer_by_kid(i) for i in range(ai.n_kmers)]
    want_tf = np.array(ai.get_tf_values(kmers), dtype=np.int64)
    top = ai.get_top_kmers(25, min_tf=2, kmer_type="23mer")
    order = [i for i in np.argsort(-want_tf, kind="stable") if want_tf[i] >= 2][:25]
    assert top == [(kmers[i], int(want_tf[i])) for i in order] and top[0][1] == int(tf.max())
    st = ai.get_kmer_frequency_stats()
    assert st["kmer_type"] == "23mer" and st["total_kmers"] == ai.n_kmers and st["max_tf"] == int(tf.max()) and st["total_tf"] == int(want_tf.sum())
    assert st["non_zero_kmers"] + st["zero_kmers"] == st["total_kmers"] and abs(st["avg_tf"] - want_tf[want_tf > 0].mean()) < 1e-9


def test_concurrent_host_calls_one_handle(canon_case, ix13):
    """ctypes drops the GIL inside the library: four threads hammer ONE 23-mer handle (tiny and large lookups, coverage,
    positions fill, counting) and one 13-mer handle (counting, lookups) plus the handle-less K1 call at the same time; every
    answer must equal the answer of the same call made alone. Exercises the per-handle staging, the scratch pool, the
    counting workspace and its cross-call ordering."""
    import threading
    from aindex_amd import counting
    ix = canon_case["ix"]
    g = canon_case["genome"]
    q_big = mixed_queries(g, 200_000, 31)
    q_small = q_big[:7].copy()
    asc = synth.genome_ascii(23, 300_000)
    reads = synth.reads_plain(46, asc, 3000, 150, rc_fraction_half=True, n_rate_ppm=1000).tobytes()
    seqs = [reads[i * 151: i * 151 + 150] for i in range(40)]
    want = {
        "big": ix.tf_ascii(q_big), "small": ix.tf_ascii(q_small), "cov": ix.coverage(seqs, 0), "pos": ix.positions_fill(reads),
        "c23": ix.count23_fixed(reads, _lib.FMT_PLAIN, _lib.CANON_TRUE_RC), "c13": ix13.count13(reads, _lib.FMT_PLAIN),
        "k1": counting.count_distinct(reads, 23, _lib.CANON_TRUE_RC, 1, fmt=_lib.FMT_PLAIN),
    }
    errors = []

    def worker(which):
        try:
            for it in range(6):
                for name in which:
                    if name == "big":
                        ok = np.array_equal(ix.tf_ascii(q_big), want["big"])
                    elif name == "small":
                        ok = all(np.array_equal(ix.tf_ascii(q_small), want["small"]) for _ in range(50))
                    elif name == "cov":
                        ok = all(np.array_equal(a, b) for a, b in zip(ix.coverage(seqs, 0), want["cov"]))
                    elif name == "pos":
                        ind, pos = ix.positions_fill(reads)
                        ok = np.array_equal(ind, want["pos"][0]) and np.array_equal(pos, want["pos"][1])
                    elif name == "c23":
                        ok = np.array_equal(ix.count23_fixed(reads, _lib.FMT_PLAIN, _lib.CANON_TRUE_RC), want["c23"])
                    elif name == "c13":
                        ok = np.array_equal(ix13.count13(reads, _lib.FMT_PLAIN), want["c13"])
                    else:
                        k, c = counting.count_distinct(reads, 23, _lib.CANON_TRUE_RC, 1, fmt=_lib.FMT_PLAIN)
                        ok = np.array_equal(k, want["k1"][0]) and np.array_equal(c, want["k1"][1])
                    if not ok:
                        errors.append((name, it))
        except Exception as e:                       # noqa: BLE001
            errors.append((which, repr(e)))

    threads = [threading.Thread(target=worker, args=(w,)) for w in (("big", "small", "cov"), ("pos", "small", "c23"), ("c13", "k1", "small"), ("k1", "pos", "big", "c13"))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_tools_cli(gold, small23_prefix, tmp_path):
    from aindex_amd import tools
    out = str(tmp_path / "o")
    os.makedirs(out)
    cwd = os.getcwd()
    os.chdir(out)
    try:
        assert tools.main(["kmer_counter", os.path.join(gold, "small23", "reads.fa"), "23", "kc.txt", "-t", "4"]) == 0
        rows = sorted(ln for ln in open("output.txt").read().split("\n") if ln)
        want = sorted(ln for ln in open(small23_prefix + ".dat").read().split("\n") if ln)
        assert rows == want
        open("keys.txt", "w").write("".join(r.split("\t")[0] + "\n" for r in open(small23_prefix + ".dat").read().split("\n") if r))
        assert tools.main(["compute_mphf_seq", "keys.txt", "t.pf"]) == 0
        assert open("t.pf", "rb").read() == open(small23_prefix + ".pf", "rb").read()
        assert tools.main(["compute_index", small23_prefix + ".dat", "t.pf", "t", "4", "0"]) == 0
        assert open("t.kmers.bin", "rb").read() == open(small23_prefix + ".kmers.bin", "rb").read()
        assert open("t.tf.bin", "rb").read() == open(small23_prefix + ".tf.bin", "rb").read()
    finally:
        os.chdir(cwd)


def test_kmer_counter_sets_golden(gold):
    from aindex_amd import counting
    fa = open(os.path.join(gold, "kmer_counter", "mixed.fa"), "rb").read()
    for k in (23, 13):
        for mc in (1, 2):
            rows = [ln.split("\t") for ln in open(os.path.join(gold, "kmer_counter", f"mixed.k{k}.m{mc}.tsv")).read().split("\n") if ln]
            want = sorted((r[0], int(r[1])) for r in rows)
            keys, counts = counting.count_distinct(fa, k, _lib.CANON_REF_X86, mc)
            got = sorted((bytes(a).decode(), int(c)) for a, c in zip(synth.decode_kmers(keys, k), counts))
            assert got == want
    # TRUE_RC mode against the oracle (semantics of tests/analyze_kmers.py)
    keys, counts = counting.count_distinct(fa, 23, _lib.CANON_TRUE_RC, 1)
    okeys, ocnt = O.count_distinct(fa, 23, 2, 1)
    assert np.array_equal(keys, okeys) and np.array_equal(counts, ocnt)
    # generic-k path (k not 13/23)
    keys, counts = counting.count_distinct(fa, 17, _lib.CANON_NONE, 1)
    okeys, ocnt = O.count_distinct(fa, 17, 0, 1)
    assert np.array_equal(keys, okeys) and np.array_equal(counts, ocnt)


# ------------------------------------------------------------------------------------------------
# fingerprint filter: identical answers on/off, also on an index whose files are inconsistent
# ------------------------------------------------------------------------------------------------
def test_fingerprint_filter_on_off(canon_case):
    ix, orc = canon_case["ix"], canon_case["orc"]
    q = mixed_queries(canon_case["genome"], 300_000, 21)
    want = orc.tf_batch(q, threads=8)
    for fast in (True, False):
        ix.set_canonical_fastpath(fast)
        for fp in (True, False):
            ix.set_fingerprint_filter(fp)
            assert np.array_equal(ix.tf_ascii(q), want), (fast, fp)
    ix.set_canonical_fastpath(True)
    ix.set_fingerprint_filter(True)


def test_inconsistent_index_files_match_reference_semantics(canon_case, tmp_path):
    """checker/tf that do not agree with the MPHF (swapped / foreign entries): same answers as the oracle."""
    prefix = canon_case["prefix"]
    checker = np.fromfile(prefix + ".kmers.bin", dtype=np.uint64)
    tf = np.fromfile(prefix + ".tf.bin", dtype=np.uint32)
    checker[[0, 1]] = checker[[1, 0]]
    foreign = int(synth.sm64(77, np.array([0], dtype=np.uint64))[0]) & (4 ** 23 - 1)
    victim = int(checker[5])
    checker[5] = foreign
    p2 = str(tmp_path / "bad")
    checker.tofile(p2 + ".kmers.bin")
    tf.tofile(p2 + ".tf.bin")
    import shutil
    shutil.copy(prefix + ".pf", p2 + ".pf")
    orc = O.OracleIndex23.from_prefix(p2)
    codes = np.array([int(checker[0]), int(checker[1]), foreign, victim] + [int(c) for c in checker[100:2100]], dtype=np.uint64)
    codes = np.concatenate([codes, synth.revcomp_codes(codes, 23)])
    q = synth.decode_kmers(codes, 23)
    with Index.open_23(p2 + ".pf", p2 + ".tf.bin", p2 + ".kmers.bin") as ix:
        want = orc.tf_batch(q)
        for fp in (True, False):
            ix.set_fingerprint_filter(fp)
            assert np.array_equal(ix.tf_ascii(q), want)
            kid, strand = ix.kid_strand_ascii(q)
            assert strand.tolist() == [orc.strand(bytes(s)) for s in q]


# ------------------------------------------------------------------------------------------------
# edge cases: empty / tiny / boundary inputs
# ------------------------------------------------------------------------------------------------
def test_edge_cases(ix23, ix13, canon_case, small23_prefix):
    orc = O.OracleIndex23.from_prefix(small23_prefix)
    assert ix23.tf_ascii(b"").shape == (0,)
    assert ix23.tf_ragged([]).shape == (0,)
    assert ix23.coverage([]) == []
    assert [c.shape[0] for c in ix23.coverage(["", "A" * 22, "A" * 23])] == [0, 0, 1]
    one = open(small23_prefix + ".dat").read().split("\t")[0]
    assert ix23.tf_ascii(one.encode()).tolist() == [orc.tf(one.encode())]
    for buf in (b"", b"A", b"ACGT" * 5, b"\n\n\n", b"ACGTACGTACGTACGTACGTACG", b"ACGTACGTACGTACGTACGTACG\n"):
        for mode in (0, 1, 2):
            assert np.array_equal(ix23.count23_fixed(buf, _lib.FMT_PLAIN, mode), orc.count23_fixed(buf, False, mode))
        ind, pos = ix23.positions_fill(buf)
        oind, opos = orc.positions(buf)
        assert np.array_equal(ind, oind) and np.array_equal(pos, opos)
    from pf13 import pf13_path
    m = O.OracleMphf(pf13_path())
    for buf in (b"", b"ACGT", b"ACGTACGTACGTA", b"ACGTACGTACGT\nA", b">h\nACGTACGTACGTA", b"@r\nACGTACGTACGTAC\n+\nIIIIIIIIIIIIII"):
        assert np.array_equal(ix13.count13(buf), O.count13(m, buf, -1)), buf
    assert ix13.tf_ragged(["", "A" * 13, "A" * 12, "A" * 14]).tolist()[0] == 0
    from aindex_amd import counting
    for buf in (b"", b">x\n", b">x\nACGT\n", b"no header ACGTACGTACGTACGTACGTACGTACGT\n"):
        keys, counts = counting.count_distinct(buf, 23, _lib.CANON_REF_X86, 1)
        okeys, ocnt = O.count_distinct(buf, 23, 1, 1)
        assert np.array_equal(keys, okeys) and np.array_equal(counts, ocnt)
    # wrapper list[str] fast path == ragged path
    from aindex_amd.wrapper import AindexWrapper
    w = AindexWrapper()
    w.load_from_prefix_23mer(small23_prefix)
    q = [one, one[:-1] + "N", "ACGT", one + "A"]
    assert w.get_tf_values(q) == [orc.tf(s.encode()) for s in q]
    assert w.get_tf_values([one, one]) == [orc.tf(one.encode())] * 2
    w.close()


# ------------------------------------------------------------------------------------------------
# device-side FASTA/FASTQ normalisation == host normalisation, byte for byte
# ------------------------------------------------------------------------------------------------
def _host_norm(buf, fmt, mode):
    from aindex_amd import counting
    return counting.normalize(buf, fmt, mode)


def test_normalisers_against_reference_written_reads_files(gold):
    """Direct pin of both normalisers (device kernels and the host routine) to REFERENCE outputs: the `.reads` files the
    compiled reference's compute_reads wrote for the same inputs (tests/golden/compute_reads/, make_golden.py) are exactly
    the PLAIN form — one sequence per line: FASTA records concatenated over their lines (compute_reads.cpp:170-213 ==
    count_kmers13.cpp:211-235), FASTQ line 4i+1 (compute_reads.cpp:118-147 == count_kmers13.cpp:240-257)."""
    import torch
    from aindex_amd import counting
    d = os.path.join(gold, "compute_reads")
    cases = [("in_test.fasta", "fasta.reads", 1), ("../count13/synth.fa", "fasta_multi.reads", 1), ("in_test_se.fastq", "se.reads", 2)]
    for src, want_name, fmt in cases:
        raw = open(os.path.join(d, src), "rb").read()
        want = open(os.path.join(d, want_name), "rb").read()
        assert counting.normalize(raw, fmt, 0) == want, src
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
        assert counting.normalize_t(t, fmt, 0).cpu().numpy().tobytes() == want, src


def test_device_normalisation_equals_host(gold):
    import torch
    from aindex_amd import counting
    cases = []
    for name in ("refdata_test.fasta", "synth.fa"):
        b = open(os.path.join(gold, "count13", name), "rb").read()
        cases += [(b, 1, 0), (b, 1, 1)]
    for name in ("refdata_test_se.fastq", "refdata_test_R1.fastq", "synth.fq"):
        cases.append((open(os.path.join(gold, "count13", name), "rb").read(), 2, 0))
    cases.append((open(os.path.join(gold, "kmer_counter", "mixed.fa"), "rb").read(), 1, 1))
    tricky = [b"", b"\n", b">", b">h", b">h\n", b">h\nAC", b"ACGT\n>h\nAC\n\nGT\r\n>x>y\nTT", b"@r\nACGT", b"@r\nACGT\n+\nIIII\n@q\n\n+\n\n@z\nGG\n+\nII",
              b"no header\nACGT\n>a\nAC\n>b\n>c\nGG\n", b"\n\n>a\n\nAC\n\n", b">a\r\nAC\r\nGT\r\n>b\r\nTT"]
    for b in tricky:
        cases += [(b, 1, 0), (b, 1, 1), (b, 2, 0)]
    # a large case crossing many 256-byte chunks with long lines and long headers
    rng = np.random.default_rng(3)
    parts = []
    for i in range(3000):
        parts.append(b">" + bytes(rng.integers(65, 91, int(rng.integers(0, 400)), dtype=np.uint8)) + (b">" if i % 97 == 0 else b"") + b"\n")
        for _ in range(int(rng.integers(0, 5))):
            parts.append(bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), int(rng.integers(0, 700)))) + (b"\r\n" if i % 5 == 0 else b"\n"))
    big = b"".join(parts)
    cases += [(big, 1, 0), (big, 1, 1), (big, 2, 0)]
    for buf, fmt, mode in cases:
        want = _host_norm(buf, fmt, mode)
        t = torch.frombuffer(bytearray(buf) if buf else bytearray(1), dtype=torch.uint8)[: len(buf)].cuda()
        got = counting.normalize_t(t, fmt, mode).cpu().numpy().tobytes()
        assert got == want, (buf[:60], fmt, mode, len(got), len(want))


# ------------------------------------------------------------------------------------------------
# full-size style checks (BASELINE config sizes scaled to seconds): oracle on all host cores + invariants
# ------------------------------------------------------------------------------------------------
def test_count13_million_reads_vs_oracle_and_atomics_path(ix13):
    import torch
    from aindex_amd import engine
    from pf13 import pf13_path
    g = engine.synth_genome_t(13, 4_000_000)
    reads = engine.synth_reads_t(14, g, 1_000_000, 150, n_rate_ppm=1000)
    got = ix13.count13_t(reads).cpu().numpy().view(np.uint64)
    os.environ["AIX_COUNT13_ATOMICS"] = "1"
    try:
        got_atomic = ix13.count13_t(reads).cpu().numpy().view(np.uint64)
    finally:
        del os.environ["AIX_COUNT13_ATOMICS"]
    assert np.array_equal(got, got_atomic)                       # partition + LDS histogram == scattered atomics
    host = reads.cpu().numpy().tobytes()
    want = O.count13(O.OracleMphf(pf13_path()), host, 0, threads=min(os.cpu_count() or 8, 64))
    assert np.array_equal(got, want)
    # invariant: total = number of 13-windows free of N in every read
    r = np.frombuffer(host, dtype=np.uint8).reshape(-1, 151)[:, :150]
    isn = (r == ord("N")).astype(np.int32)
    c = np.cumsum(np.concatenate([np.zeros((r.shape[0], 1), np.int32), isn], axis=1), axis=1)
    valid = int(((c[:, 13:] - c[:, :-13]) == 0).sum())
    assert int(got.sum()) == valid


def test_count23_sum_equals_valid_windows_at_scale(canon_case):
    import torch
    from aindex_amd import engine, counting
    ix = canon_case["ix"]
    g = engine.synth_genome_t(23, 300_000)                     # the genome the index was built from
    reads = engine.synth_reads_t(41, g, 500_000, 150, rc_half=True, n_rate_ppm=1000)
    tf = ix.count23_fixed_t(reads, _lib.CANON_TRUE_RC)
    codes = counting.window_codes_t(reads, 23, _lib.CANON_TRUE_RC)
    valid = int((codes != -1).sum().item())
    assert int(tf.to(torch.int64).sum().item()) == valid        # every N-free window of a genome read is a key
    # and the per-key histogram equals the sort/run-length result scattered through the MPHF
    keys, counts = counting.count_distinct_t(reads, 23, _lib.CANON_TRUE_RC)
    exp = np.zeros(ix.n, dtype=np.int64)
    slot = np.searchsorted(canon_case["keys"], keys.cpu().numpy().view(np.uint64))
    assert np.array_equal(canon_case["keys"][slot], keys.cpu().numpy().view(np.uint64))
    # slot in sorted-key order -> MPHF slot via kid lookup of the keys
    kid, strand = ix.kid_strand_ascii(synth.decode_kmers(keys.cpu().numpy().view(np.uint64), 23))
    assert (strand == 1).all()
    exp[kid.astype(np.int64)] = counts.cpu().numpy()
    assert np.array_equal(tf.cpu().numpy().astype(np.int64), exp)


# ------------------------------------------------------------------------------------------------
# BASELINE.json's full sizes, through size-independent properties (the oracle cannot run these in seconds)
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config3_index():
    """The index of BASELINE configs[2..4]: ~5e7 true-canonical 23-mers of a synthetic 50 Mbp genome (seed 23), tf = genome
    multiplicity; keys / counts by the GPU counter, MPHF by the GPU builder, scatter on the device — no host step."""
    import torch
    from aindex_amd import engine, counting
    g = engine.synth_genome_t(23, 50_000_000)
    keys, counts = counting.count_distinct_t(g, 23, _lib.CANON_TRUE_RC)
    pf = builder.build_pf_codes_t(keys, 23)
    ix = Index.build_23_codes_t(pf, keys, counts.to(torch.int32))
    yield {"ix": ix, "g": g, "keys": keys, "counts": counts, "pf": pf}
    ix.close()


def test_config4_share_of_25M_reads_properties(config3_index):
    """Config 4 at the N = 8 share (25 M reads x 150 bp, seed 41, 50 % reverse strand, 0.1 % N): (i) sum(tf) == the number of
    windows without N (every such window of a genome read is a key), counted independently with torch; (ii) the histogram is
    the sort / run-length result of the same reads (multiset of counts, and key by key on a sample through kid lookups);
    (iii) the atomics back end gives the same table."""
    import torch
    from aindex_amd import engine, counting
    ix, g = config3_index["ix"], config3_index["g"]
    n_reads = 25_000_000
    reads = engine.synth_reads_t(41, g, n_reads, 150, rc_half=True, n_rate_ppm=1000)
    tf = ix.count23_fixed_t(reads, _lib.CANON_TRUE_RC)
    torch.cuda.synchronize()
    is_n = (reads.view(n_reads, 151)[:, :150] == ord("N")).to(torch.int16)
    c = torch.cumsum(is_n, dim=1)
    inside = c[:, 22:] - torch.cat([torch.zeros(n_reads, 1, dtype=torch.int16, device=c.device), c[:, :-23]], dim=1)
    valid = int((inside == 0).sum().item())
    del is_n, c, inside
    assert int(tf.to(torch.int64).sum().item()) == valid and valid > n_reads * 100
    # K1 on ALL the reads (three pieces merged): every distinct k-mer it reports is a key of the index, and its counts are the histogram's
    # (a chunk handed to the neighbouring partition shows up as keys the index does not hold: the round-3 search bug, found at 200 M reads)
    k_all, c_all = counting.count_distinct_t(reads, 23, _lib.CANON_TRUE_RC)
    assert bool((ix.tf_codes_t(k_all) != 0).all()) and bool((k_all[1:] > k_all[:-1]).all())
    tfl = tf.to(torch.int64)
    assert torch.equal(torch.sort(tfl[tfl > 0]).values, torch.sort(c_all.to(torch.int64)).values)
    del k_all, c_all, tfl
    keys, counts = counting.count_distinct_t(reads[: 5_000_000 * 151], 23, _lib.CANON_TRUE_RC)       # sort / run-length of the first 5 M reads
    tf5 = ix.count23_fixed_t(reads[: 5_000_000 * 151], _lib.CANON_TRUE_RC).to(torch.int64)
    assert torch.equal(torch.sort(tf5[tf5 > 0]).values, torch.sort(counts).values)
    pick = torch.arange(0, keys.numel(), max(1, keys.numel() // 200_000), device=keys.device)
    kid, strand = ix.kid_strand_ascii(synth.decode_kmers(keys[pick].cpu().numpy().view(np.uint64), 23))
    assert (strand == 1).all()
    assert np.array_equal(tf5.cpu().numpy()[kid.astype(np.int64)], counts[pick].cpu().numpy())
    # the three back ends give one histogram: the probe + LDS-histogram path and the "distinct k-mers first" path forced in turn (25 M reads against
    # 5e7 keys sit right at the 64-windows-per-key threshold of the automatic choice), the atomics path below
    for force, backend in (("1", 3), ("0", 2)):
        os.environ["AIX_COUNT23_VIA_K1"] = force
        try:
            tf_k = ix.count23_fixed_t(reads, _lib.CANON_TRUE_RC)
            assert ix.info["count23_backend"] == backend
        finally:
            del os.environ["AIX_COUNT23_VIA_K1"]
        assert torch.equal(tf_k, tf)
        del tf_k
    os.environ["AIX_COUNT23_ATOMICS"] = "1"
    try:
        tf_a = ix.count23_fixed_t(reads, _lib.CANON_TRUE_RC)
    finally:
        del os.environ["AIX_COUNT23_ATOMICS"]
    assert torch.equal(tf_a, tf)


def test_config3_full_size_index_against_the_oracle(config3_index, tmp_path):
    """The 5e7-key index of configs[2] itself against the oracle (until now it was only checked through properties and
    HIP-vs-HIP): (a) every stored key, forward and reverse-complemented, through the codes entry point == its count; (b) 2e7 Q_mix + 2e7
    Q_rand (the bench's generators) with N / lower-case sprinkled in, against the oracle's get_tf_values on all host cores; (c) the same
    with the verification table off and with the absence filter off; (d) total / both / kid / strand / hash on 1e6 of them."""
    import torch
    from aindex_amd import engine
    ix, g, keys, counts = config3_index["ix"], config3_index["g"], config3_index["keys"], config3_index["counts"]
    n = keys.numel()
    assert 49_000_000 < n < 51_000_000 and ix.canonical_only
    # (a) all keys, both strands
    M = (1 << 46) - 1
    x = (~keys) & M

    def swap(v, m, s):
        return ((v >> s) & m) | ((v & m) << s)
    x = swap(x, 0x3333333333333333, 2)
    x = swap(x, 0x0F0F0F0F0F0F0F0F, 4)
    x = swap(x, 0x00FF00FF00FF00FF, 8)
    x = swap(x, 0x0000FFFF0000FFFF, 16)
    x = swap(x, 0x00000000FFFFFFFF, 32)
    rc = (x >> 18) & M
    want32 = counts.to(torch.int32)
    assert torch.equal(ix.tf_codes_t(keys), want32)
    assert torch.equal(ix.tf_codes_t(rc), want32)
    assert bool((rc >= keys).all())                                         # true-canonical key set
    del x, rc
    # the oracle on the same index files (written from the handle: what load_hash would read, hash.cpp:367-450)
    prefix = str(tmp_path / "c3")
    open(prefix + ".pf", "wb").write(config3_index["pf"])                    # the image the handle was built from (GPU builder)
    ix.tf_array().tofile(prefix + ".tf.bin")
    ix.checker_array().tofile(prefix + ".kmers.bin")
    orc = O.OracleIndex23.from_prefix(prefix)
    assert orc.n == n
    # (b) queries
    nq = 20_000_000
    q = torch.cat([engine.synth_mix23_t(8, g, nq), engine.synth_kmers_t(7, nq, 23)]).cpu().numpy().reshape(-1, 23).copy()
    idx = np.arange(0, q.shape[0], 97)
    q[idx, (idx * 7) % 23] = ord("N")
    idx = np.arange(5, q.shape[0], 131)
    q[idx] |= 0x20
    flatq = q.reshape(-1)
    want = orc.tf_batch(flatq, threads=max(1, min(64, os.cpu_count() or 1)))
    assert int((want != 0).sum()) > nq // 3
    assert np.array_equal(ix.tf_ascii(flatq), want)
    # (c) table off / filter off
    try:
        ix.set_bucket_table(False)
        assert np.array_equal(ix.tf_ascii(flatq), want)
        ix.set_bucket_table(True)
        ix.set_absence_filter(False)
        assert np.array_equal(ix.tf_ascii(flatq), want)
    finally:
        ix.set_bucket_table(True)
        ix.set_absence_filter(True)
    # (d) the other query kinds on 1e6 (every 40th query: both generators, sprinkles included)
    sub = np.ascontiguousarray(q[::40]).reshape(-1)
    okid, ostrand, ototal, ofwd, orc_ = orc.info_batch(sub)
    kid, strand = ix.kid_strand_ascii(sub)
    assert np.array_equal(kid, okid) and np.array_equal(strand, ostrand)
    assert np.array_equal(ix.total_ascii(sub), ototal)
    f, r = ix.both_ascii(sub)
    assert np.array_equal(f, ofwd.astype(np.uint64)) and np.array_equal(r, orc_.astype(np.uint64))
    assert np.array_equal(ix.hash_ascii(sub), orc.hash_batch(sub))
    assert np.array_equal(ix.tf_ascii(sub), want[::40])


def test_wave_lower_bound_of_the_partition_kernels():
    """The wave-wide search with which the histogram / MSD kernels find a partition's chunks in the sorted directory (32 probes per round), against
    numpy.searchsorted: every key of directories whose sizes make a round's span a multiple of 33 (where the first version, for which every probe of
    a round said "less", cut the upper bound by one and handed a chunk to the next partition: 371 of 2.5e10 windows at config-4 size), sizes 0 and 1,
    few and many distinct values."""
    import torch
    from aindex_amd._lib import lib, check, vp
    rng = np.random.default_rng(33)
    sizes = [0, 1, 2, 31, 32, 33, 34, 65, 66, 33 * 33, 33 * 33 + 1, 33 * 34, 33 ** 3, 33 ** 3 - 1, 2 * 33 ** 3, 3042, 2202, 1_000_003, 4_000_000] + [int(x) for x in rng.integers(2, 300_000, size=25)]
    for n in sizes:
        for nv in (1, 3, 2049):
            a = np.sort(rng.integers(0, nv, size=n)).astype(np.uint16)
            keys = np.arange(0, min(nv, 2049) + 2, dtype=np.uint32) if nv < 100 else np.concatenate([rng.integers(0, 2050, size=200), [0, 2047, 2048, 2049]]).astype(np.uint32)
            da = torch.from_numpy(a.view(np.int16)).cuda() if n else torch.zeros(1, dtype=torch.int16, device="cuda")
            dk = torch.from_numpy(keys.view(np.int32)).cuda()
            out = torch.empty(2 * keys.shape[0], dtype=torch.int32, device="cuda")
            check(lib().aix_selftest_lower_bound_dev(vp(da.data_ptr()), n, vp(dk.data_ptr()), keys.shape[0], vp(out.data_ptr()), None), "selftest_lower_bound")
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint32).reshape(-1, 2)
            want = np.stack([np.searchsorted(a, keys, side="left"), np.searchsorted(a, keys + 1, side="left")], axis=1).astype(np.uint32)
            assert np.array_equal(got, want), (n, nv, keys[np.nonzero((got != want).any(axis=1))[0][:4]])


def test_config5_coverage_full_size_properties(config3_index, ix13):
    """Config 5 at its FULL size (1 M sequences x 10 kbp, as the bench runs it; round 2 tested a tenth): the per-position profile equals
    the batch lookup of every window, checked on sampled sequences, for k = 23 (index of config 3) and k = 13. 10 GB of sequences and a
    40 GB profile per k live in HBM for the duration of the check."""
    import torch
    from aindex_amd import engine
    from pf13 import pf13_path
    ix, g = config3_index["ix"], config3_index["g"]
    n_seq, L = 1_000_000, 10_000
    g13 = engine.synth_genome_t(13, 4_000_000)
    own13 = Index.open_13(pf13_path(), None)                                         # its table = the 13-mer counts of g13: every window of a forward read answers >= 1
    own13.set_tf_13(ix13.count13_t(g13).cpu().numpy().view(np.uint64))
    for index, k, genome in ((ix, 23, g), (own13, 13, g13)):
        seqs = engine.synth_reads_t(51, genome, n_seq, L, rc_half=(k == 23), n_rate_ppm=1000)
        offs = torch.arange(0, (n_seq + 1) * (L + 1), L + 1, dtype=torch.int64, device="cuda")
        per = (L + 1) - k + 1
        ooffs = torch.arange(0, (n_seq + 1) * per, per, dtype=torch.int64, device="cuda")
        prof = index.coverage_t(seqs, offs, ooffs, n_seq * per, 0).view(n_seq, per)
        torch.cuda.synchronize()
        # (the last window of a record reaches into the '\n': for k = 23 it is answered like any query with a foreign byte — the
        # reverse-complement probe of its sanitised code may hit, python_wrapper.cpp:610-627 — so it is not asserted to be 0)
        rows = seqs.view(n_seq, L + 1)
        # (sampled on both sides of every multiple of 2^31 bytes of the batch: rows 214 726, 429 453, 644 180 and 858 907 hold those positions)
        for sidx in (0, 1, 4_999, 50_000, 99_999, 214_725, 214_726, 214_727, 300_000, 429_452, 429_453, 429_454, 500_000, 644_179, 644_180, 644_181, 700_000,
                     858_906, 858_907, 858_908, 999_999):
            win = rows[sidx].unfold(0, k, 1).contiguous().view(-1)                   # every window of the record (incl. the one over the '\n'), as a query batch
            want = index.tf_ascii_t(win)
            assert torch.equal(prof[sidx], want), (k, sidx)
        # every record answers like every other one: windows without N are keys (k = 23: genome reads; k = 13: nearly all 13-mers occur in 4 Mbp),
        # so no stretch of records may come back empty — the sampled rows above cannot see that, this can (it found rounds 1-2's sign extension)
        rownz = torch.empty(n_seq, dtype=torch.int64, device="cuda")
        for lo in range(0, n_seq, 50_000):                                           # in slabs: a boolean copy of the whole profile would be another 10 GB
            rownz[lo:lo + 50_000] = (prof[lo:lo + 50_000, : L - k + 1] != 0).sum(dim=1)
        if k == 23:
            assert int(rownz.min().item()) > 0.9 * (L - k + 1), int(rownz.argmin().item())
            assert int(rownz.sum().item()) / (n_seq * (L - k + 1)) > 0.97          # (1 - 0.001)^23 = 0.977 of the windows hold no N
        else:
            assert int(rownz.min().item()) > 0.9 * (L - k + 1), int(rownz.argmin().item())
        del rownz
        del prof, seqs, rows, offs, ooffs
        torch.cuda.empty_cache()
    own13.close()


# ------------------------------------------------------------------------------------------------
# GPU MWHC builder: a valid emphf .pf (not byte-identical to the reference's), usable end to end
# ------------------------------------------------------------------------------------------------
def test_gpu_builder_makes_valid_reference_compatible_pf(tmp_path):
    import torch
    from aindex_amd import engine, counting
    g = engine.synth_genome_t(99, 400_000)
    keys_t, counts_t = counting.count_distinct_t(g, 23, _lib.CANON_TRUE_RC)
    pf = builder.build_pf_codes_t(keys_t, 23)
    keys = keys_t.cpu().numpy().view(np.uint64)
    host_pf = builder.build_pf_codes(keys, 23)
    assert len(pf) == len(host_pf) and pf[:32] == host_pf[:32]          # same n, hash domain, seed, bit-pair count
    path = str(tmp_path / "g.pf")
    open(path, "wb").write(pf)
    m = O.OracleMphf(path)                                              # the reference's evaluator (restated) accepts it
    kmers = synth.decode_kmers(keys, 23)
    slots = O.lib()  # noqa: F841
    h = np.array([m.lookup(bytes(k)) for k in kmers[:20000]])
    assert len(set(h.tolist())) == 20000 and h.max() < keys.shape[0]
    # full bijection through the GPU evaluator, then the whole query path against the oracle on the same files
    ix = Index.build_23_codes_t(pf, keys_t, counts_t.to(torch.int32))
    slots_gpu = ix.hash_ascii(kmers)
    assert np.array_equal(np.sort(slots_gpu), np.arange(keys.shape[0], dtype=np.uint64))
    prefix = str(tmp_path / "g")
    ix.checker_array().tofile(prefix + ".kmers.bin")
    ix.tf_array().tofile(prefix + ".tf.bin")
    orc = O.OracleIndex23.from_prefix(prefix)
    q = np.concatenate([kmers[::7], synth.decode_kmers(synth.revcomp_codes(keys[::11], 23), 23), synth.random_kmers_ascii(5, 20000, 23)])
    assert np.array_equal(ix.tf_ascii(q), orc.tf_batch(q, threads=8))
    ref_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    try:
        import sys
        sys.path.insert(0, ref_dir)
        import aindex_cpp                                               # the compiled reference, when it travelled with the repo
    except Exception:
        aindex_cpp = None
    if aindex_cpp is not None:
        w = aindex_cpp.AindexWrapper()
        w.load_from_prefix_23mer(prefix)
        sub = q[:3000]
        assert list(w.get_tf_values([bytes(s).decode() for s in sub])) == ix.tf_ascii(sub).tolist()
    ix.close()


# ------------------------------------------------------------------------------------------------
# BASELINE config 1 plumbing through the mirrors: count_kmers13 -> AIndex.load_from_prefix(13) -> get_tf_values
# ------------------------------------------------------------------------------------------------
def test_config1_plumbing_13mer_through_mirrors(gold, tmp_path):
    import shutil
    from pf13 import pf13_path
    from aindex_amd import tools
    from aindex_amd.aindex import AIndex
    prefix = str(tmp_path / "t13")
    shutil.copy(pf13_path(), prefix + ".pf")
    fa = os.path.join(gold, "count13", "refdata_test.fasta")
    assert tools.main(["count_kmers13", fa, prefix + ".pf", prefix + ".tf.bin", "4"]) == 0
    z = np.load(os.path.join(gold, "count13", "expected.npz"))
    tf = np.fromfile(prefix + ".tf.bin", dtype=np.uint64)
    assert tf.shape[0] == 4 ** 13 and np.array_equal(np.nonzero(tf)[0].astype(np.uint64), z["refdata_test.fasta.idx"])
    ai = AIndex.load_from_prefix(prefix, kmer_size=13)
    seqs = [ln for ln in open(fa).read().split("\n") if ln and not ln.startswith(">")]
    wins = [s[i:i + 13] for s in seqs for i in range(len(s) - 12)]
    got = ai.get_tf_values(wins)
    assert len(wins) == 12 and got == [1] * 12                       # "twelve 1s"
    assert ai.get_tf_value(wins[0]) == 1 and ai.get_tf_value(wins[0].lower()) == 0
    assert ai.get_total_tf_value_13mer(wins[0]) >= 1
    assert ai.get_sequence_coverage(seqs[0], 0, 13) == [1] * (len(seqs[0]) - 12)
    assert ai._wrapper.get_tf_by_index_13mer(int(z["refdata_test.fasta.idx"][0])) == 1
    assert len(ai) == 4 ** 13 and ai.n_kmers == 4 ** 13              # get_hash_size(): TOTAL_13MERS in 13-mer mode (python_wrapper.cpp:846-851)


def test_error_codes(small23_prefix, tmp_path):
    import ctypes as C
    L = _lib.lib()
    h = C.c_void_p()
    assert L.aix_index_open_23(b"/nonexistent.pf", b"/x", b"/y", 0, C.byref(h)) == -2          # AIX_ERR_IO
    bad = str(tmp_path / "bad.pf")
    open(bad, "wb").write(open(small23_prefix + ".pf", "rb").read()[:100])
    assert L.aix_index_open_23(bad.encode(), (small23_prefix + ".tf.bin").encode(), (small23_prefix + ".kmers.bin").encode(), 0, C.byref(h)) == -3
    with pytest.raises(FileNotFoundError):
        from aindex_amd.wrapper import AindexWrapper
        AindexWrapper().load_from_prefix_23mer("/nonexistent/prefix")
    with Index.open_23(small23_prefix + ".pf", small23_prefix + ".tf.bin", small23_prefix + ".kmers.bin") as ix:
        out = np.zeros(4 ** 13, dtype=np.uint64)
        assert L.aix_count13(ix._h, None, 0, 0, out.ctypes.data_as(C.c_void_p)) == -7           # AIX_ERR_MODE
        assert L.aix_tf_batch_ascii(ix._h, None, 5, None) == -1                                  # AIX_ERR_ARG
    assert L.aix_index_open_23(small23_prefix.encode() + b".pf", (small23_prefix + ".tf.bin").encode(), (small23_prefix + ".kmers.bin").encode(), 99,
                               C.byref(h)) == -1                                                 # no such device


@pytest.mark.parametrize("piece", [1, 13, 1000, 77_777, 1 << 20])
def test_count13_in_pieces(ix13, piece):
    """Buffers beyond 2^31 windows are counted piece by piece (12-byte overlap, later pieces add to the table).
    AIX_COUNT13_PIECE shrinks the piece so the cut logic runs at test sizes; every cut position must give the
    one-piece result (cuts fall inside reads, on separators, on N)."""
    import torch
    from aindex_amd import engine
    g = engine.synth_genome_t(13, 200_000)
    reads = engine.synth_reads_t(15, g, 3000 if piece >= 1000 else 40, 150, n_rate_ppm=3000)
    want = ix13.count13_t(reads).cpu().numpy().view(np.uint64)
    os.environ["AIX_COUNT13_PIECE"] = str(piece)
    try:
        got = ix13.count13_t(reads).cpu().numpy().view(np.uint64)
    finally:
        del os.environ["AIX_COUNT13_PIECE"]
    assert want.sum() > 0 and np.array_equal(got, want)


def test_count13_two_streams_share_the_workspace(ix13):
    """Counting calls on one handle share its partition workspace: issued back to back on two streams they must still
    produce the results of two separate runs (the second call waits on an event recorded behind the first)."""
    import torch
    from aindex_amd import engine
    g = engine.synth_genome_t(13, 1_000_000)
    ra = engine.synth_reads_t(17, g, 600_000, 150, n_rate_ppm=1000)
    rb = engine.synth_reads_t(18, g, 500_000, 150, n_rate_ppm=5000)
    want_a, want_b = ix13.count13_t(ra).clone(), ix13.count13_t(rb).clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            got_a = ix13.count13_t(ra)
        with torch.cuda.stream(s2):
            got_b = ix13.count13_t(rb)
        with torch.cuda.stream(s1):
            got_a2 = ix13.count13_t(ra)
        torch.cuda.synchronize()
        assert bool(torch.equal(got_a, want_a)) and bool(torch.equal(got_b, want_b)) and bool(torch.equal(got_a2, want_a))
    assert not bool(torch.equal(want_a, want_b))


@pytest.mark.slow
def test_count13_beyond_4gib(ix13):
    """30 M reads = 4.53 GB > 2^32 bytes: three pieces through the partitioned path == scattered atomics; the total is
    the number of N-free windows (computed from a second, independent kernel: the window-code kernel)."""
    import torch
    from aindex_amd import counting, engine
    g = engine.synth_genome_t(13, 4_000_000)
    reads = engine.synth_reads_t(16, g, 30_000_000, 150, n_rate_ppm=1000)
    assert reads.numel() > (1 << 32)
    got = ix13.count13_t(reads)
    os.environ["AIX_COUNT13_ATOMICS"] = "1"
    try:
        ref = ix13.count13_t(reads)
    finally:
        del os.environ["AIX_COUNT13_ATOMICS"]
    assert bool(torch.equal(got, ref))
    total = 0
    step = 1 << 30
    for lo in range(0, reads.numel() - 12, step):
        c = counting.window_codes_t(reads[lo: lo + step + 12], 13, _lib.CANON_NONE)
        total += int((c != -1).sum().item())
        del c
    assert int(got.sum().item()) == total


def test_count13_skewed_and_crlf_inputs(ix13):
    """Extreme partition skew (homopolymers, short tandem repeats) and CRLF / mixed-case records."""
    from pf13 import pf13_path
    m = O.OracleMphf(pf13_path())
    rng = np.random.default_rng(11)
    lines = []
    for i in range(4000):
        kind = i % 5
        if kind == 0:
            lines.append(b"A" * 150)
        elif kind == 1:
            lines.append(b"T" * 70 + b"N" + b"T" * 79)
        elif kind == 2:
            lines.append(b"ACG" * 50)
        elif kind == 3:
            lines.append(bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 150)))
        else:
            lines.append(bytes(rng.choice(np.frombuffer(b"acgtACGTnRY", dtype=np.uint8), 150)))
    plain = b"\n".join(lines) + b"\n"
    got = ix13.count13(plain, _lib.FMT_PLAIN)
    assert np.array_equal(got, O.count13(m, plain, 0, threads=8))
    assert int(got.max()) >= 800 * 138                                 # the poly-A bin
    fq = b"".join(b"@r%d\r\n" % i + ln + b"\r\n+\r\n" + b"I" * len(ln) + b"\r\n" for i, ln in enumerate(lines[:500]))
    assert np.array_equal(ix13.count13(fq), O.count13(m, fq, -1))      # '\r' stays in the sequence line -> breaks windows
    fa = b"".join(b">s%d\r\n" % i + ln[:75] + b"\r\n" + ln[75:] + b"\r\n" for i, ln in enumerate(lines[:500]))
    assert np.array_equal(ix13.count13(fa), O.count13(m, fa, -1))


def test_count13_region_guard_and_foreign_pf(ix13, tmp_path, monkeypatch):
    """(i) The chunk-region bound of the partitioned counter is no longer a silent guard: with an undersized region
    (AIX_COUNT13_TEST_REGION) the call fails with AIX_ERR_UNSUPPORTED instead of returning short counts. (ii) A 13-mer handle
    opened on a .pf that is NOT the all-13-mers one (code -> slot is not a bijection: slots collide and overflow 4^13) counts
    like the reference, counts[mphf(window)] += 1 for slots < 4^13 (count_kmers13.cpp:147-152), through the adding path."""
    asc = synth.genome_ascii(5, 60_000)
    reads = synth.reads_plain(6, asc, 3000, 150, n_rate_ppm=1000).tobytes()
    monkeypatch.setenv("AIX_COUNT13_TEST_REGION", "100")
    with pytest.raises(_lib.AixError) as ei:
        ix13.count13(reads, _lib.FMT_PLAIN)
    assert ei.value.status == -6
    monkeypatch.delenv("AIX_COUNT13_TEST_REGION")
    from pf13 import pf13_path
    want = O.count13(O.OracleMphf(pf13_path()), reads, 0)
    assert np.array_equal(ix13.count13(reads, _lib.FMT_PLAIN), want)                 # the handle is fine afterwards
    rng = np.random.default_rng(11)
    keys = sorted({bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 13)) for _ in range(6000)})
    pf = builder.build_pf(keys)
    path = str(tmp_path / "foreign13.pf")
    open(path, "wb").write(pf)
    m = O.OracleMphf(path)
    with Index.create_13(pf, None) as fx:
        got = fx.count13(reads, _lib.FMT_PLAIN)
        want = O.count13(m, reads, 0)
        assert np.array_equal(got, want) and int(got.sum()) > 0 and int(got.max()) > 100 and int(np.count_nonzero(got)) <= 6001   # ~6 000 slots shared by every window
        q = b"".join(keys[:50]) + reads[:13]
        fx.set_tf_13(got)
        assert fx.tf_ascii(q).tolist() == [int(want[m.lookup(q[i:i + 13])]) & 0xFFFFFFFF if m.lookup(q[i:i + 13]) < 4 ** 13 else 0 for i in range(0, len(q), 13)]


def test_count_workspace_that_does_not_fit_halves_the_pass(small23_prefix, monkeypatch):
    """The partition workspace of count13 / count23 is sized by the windows of one pass (2^30 windows: ~10.6 GiB). When it does not fit
    (AIX_COUNT_TEST_WORKSPACE_MAX plays the full device) the pass is halved until it does, with the same counts; a limit nothing
    fits under fails with AIX_ERR_NOMEM and leaves the handle usable."""
    from pf13 import pf13_path
    asc = synth.genome_ascii(15, 80_000)
    reads = synth.reads_plain(16, asc, 4000, 150, n_rate_ppm=500).tobytes()
    want13 = O.count13(O.OracleMphf(pf13_path()), reads, 0)
    gold = open(small23_prefix + ".reads", "rb").read().replace(b"~", b"\n") * 10          # ~600 K windows: 19 tiles of the split kernel
    want23 = O.OracleIndex23.from_prefix(small23_prefix).count23_fixed(gold, False, 1)
    monkeypatch.setenv("AIX_COUNT23_HIST_MIN", "0")
    for kind in ("13", "23"):
        def run(limit):
            if limit is None:
                monkeypatch.delenv("AIX_COUNT_TEST_WORKSPACE_MAX", raising=False)
            else:
                monkeypatch.setenv("AIX_COUNT_TEST_WORKSPACE_MAX", str(limit))
            ix = Index.open_13(pf13_path(), None) if kind == "13" else Index.open_23(small23_prefix + ".pf", small23_prefix + ".tf.bin", small23_prefix + ".kmers.bin")
            try:
                before = ix.info["device_bytes"]
                got = ix.count13(reads, _lib.FMT_PLAIN) if kind == "13" else ix.count23_fixed(gold, _lib.FMT_PLAIN, 1)
                return got, ix.info["device_bytes"] - before
            finally:
                ix.close()
        want = want13 if kind == "13" else want23
        got, full = run(None)
        assert np.array_equal(got, want) and full > 0
        got, half = run(full // 2)                                     # at least one halving
        assert np.array_equal(got, want) and 0 < half <= full // 2
        got, small = run(full // 5)
        assert np.array_equal(got, want) and 0 < small <= full // 5
        monkeypatch.setenv("AIX_COUNT_TEST_WORKSPACE_MAX", "1")
        ix = Index.open_13(pf13_path(), None) if kind == "13" else Index.open_23(small23_prefix + ".pf", small23_prefix + ".tf.bin", small23_prefix + ".kmers.bin")
        with pytest.raises(_lib.AixError) as ei:
            ix.count13(reads, _lib.FMT_PLAIN) if kind == "13" else ix.count23_fixed(gold, _lib.FMT_PLAIN, 1)
        assert ei.value.status == -4
        monkeypatch.delenv("AIX_COUNT_TEST_WORKSPACE_MAX")
        got = ix.count13(reads, _lib.FMT_PLAIN) if kind == "13" else ix.count23_fixed(gold, _lib.FMT_PLAIN, 1)
        assert np.array_equal(got, want)
        ix.close()


def test_early_exit_walk_on_off(canon_case, ix23, q23):
    """Presence-mask early exit: identical answers in every combination of the three switches, on the canonical
    synthetic index and on the reference-built (non-canonical) golden index."""
    ix, orc = canon_case["ix"], canon_case["orc"]
    q = mixed_queries(canon_case["genome"], 300_000, 33)
    want = orc.tf_batch(q, threads=8)
    for fast in (True, False):
        for fp in (True, False):
            for ee in (True, False):
                for bk in (True, False):
                    ix.set_canonical_fastpath(fast); ix.set_fingerprint_filter(fp); ix.set_early_exit(ee); ix.set_bucket_table(bk)
                    assert np.array_equal(ix.tf_ascii(q), want), (fast, fp, ee, bk)
    ix.set_canonical_fastpath(True); ix.set_fingerprint_filter(True); ix.set_early_exit(True); ix.set_bucket_table(False)
    kid, strand = ix.kid_strand_ascii(q[:5000])
    assert strand.tolist() == [orc.strand(bytes(s)) for s in q[:5000]]
    qs = flat(q23["queries"])
    for ee in (True, False):
        ix23.set_early_exit(ee)
        assert ix23.tf_ascii(qs).tolist() == q23["tf"]
        assert ix23.total_ascii(qs).tolist() == q23["total"]
    ix23.set_early_exit(True)
    import torch
    from aindex_amd import engine
    li = ix.lines_ascii_t(engine.synth_kmers_t(7, 200_000, 23)).cpu().numpy()
    assert 1.0 <= float((li & 15).mean()) < 2.0          # MPHF path: absent keys stop after ~1.35 records on average
    ix.set_bucket_table(True)
    ix.set_absence_filter(False)
    li = ix.lines_ascii_t(engine.synth_kmers_t(7, 200_000, 23)).cpu().numpy()
    assert float((li >> 16).mean()) == 1.0 and float((li & 15).mean()) < 0.2   # table on: one bucket line; the MPHF only behind an overflowed bucket
    ix.set_absence_filter(True)
    li = ix.lines_ascii_t(engine.synth_kmers_t(7, 200_000, 23)).cpu().numpy()
    assert float(((li >> 12) & 15).mean()) == 1.0 and float((li >> 16).mean()) < 0.05   # filter on: one cached word, absent keys rarely reach the table


def test_positions13_reference_pin_oracle_pieces_and_mirror(gold, tmp_path, monkeypatch):
    """N3 (compute_aindex13): (a) pinned to the reference — the tool indexes the tf file misread as u32 (compute_aindex13.cpp:46-47);
    a handle holding that view (widened to u64) must reproduce the reference's 1-thread .index.bin / .indices.bin
    (tests/golden/aindex13); (b) with the real u64 table: equal to the oracle restatement, in one piece, cut into pieces, and through
    the device-resident twin; (c) the compute_aindex13 shim writes those files and the AindexWrapper mirror answers
    get_positions_13mer from them (the reference never maps its positions file, python_wrapper.cpp:439-471)."""
    import hashlib
    import subprocess
    import torch
    from pf13 import pf13_path
    from test_oracle_golden13 import _misread_tf_view
    g = np.load(os.path.join(gold, "aindex13", "synth.npz"))
    reads = open(os.path.join(gold, "count13", "synth.txt"), "rb").read()
    m = O.OracleMphf(pf13_path())
    with Index.open_13(pf13_path(), None) as ix:
        ix.set_tf_13(_misread_tf_view(gold))                                        # (a)
        indices, pos = ix.positions_fill(reads)
        assert np.array_equal(pos, g["positions"]) and int((pos != 0).sum()) > 0
        assert hashlib.sha256(indices.tobytes()).digest() == bytes(g["indices_sha256"])
        tf = ix.count13(reads, _lib.FMT_PLAIN)                                      # (b) the table count_kmers13 writes for these reads
        ix.set_tf_13(tf)
        noisy = reads[:40_000] + b"acgtacgtacgtacgtacgt\nACGTNACGTACGTACGTRACGT~ACGTACGTACGTACGTAC?GGGGGGGGGGGGGGGG\n" + reads[40_000:90_000]
        for buf in (reads, noisy, b"?AC\n" + reads[:5000], b"ACGTACGTACGTA", b"ACGT", b""):
            oind, opos = O.positions13(m, tf, buf)
            ind, pos = ix.positions_fill(buf)
            assert np.array_equal(ind, oind) and np.array_equal(pos, opos), len(buf)
            for piece in ("1000", "77777"):
                monkeypatch.setenv("AIX_POSITIONS_PIECE", piece)
                ind2, pos2 = ix.positions_fill(buf)
                monkeypatch.delenv("AIX_POSITIONS_PIECE")
                assert np.array_equal(pos2, opos), (len(buf), piece)
            # the MSD grouping (the default from 2^22 windows up) on a 13-mer handle: 4^13 slots, u64 tf, forward-strand buckets; also
            # with buckets set aside and cut into pieces
            for env in ({"AIX_A2_MSD": "1"}, {"AIX_A2_MSD": "1", "AIX_A2_TEST_CAP": "5", "AIX_POSITIONS_PIECE": "50000"}):
                for k_, v_ in env.items():
                    monkeypatch.setenv(k_, v_)
                ind3, pos3 = ix.positions_fill(buf)
                for k_ in env:
                    monkeypatch.delenv(k_)
                assert np.array_equal(ind3, oind) and np.array_equal(pos3, opos), (len(buf), env)
            if buf:
                t = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
                indt, post = ix.positions_fill_t(t)
                assert np.array_equal(indt.cpu().numpy().view(np.uint64), oind) and np.array_equal(post.cpu().numpy().view(np.uint64), opos)
        assert int((opos != 0).sum()) == 0                                          # (the empty buffer came last)
        oind, opos = O.positions13(m, tf, reads)
        assert 0 < int((opos != 0).sum()) <= int(tf.sum())                          # count_kmers13 upper-cases its input, compute_aindex13 skips lower-case windows
        # hasher_13mer.lookup through the ABI == the oracle's MPHF
        ks = [reads[i:i + 13] for i in (0, 7, 150, 2000)] + [b"ACGTNACGTACGT"]
        assert ix.hash_ascii(b"".join(ks)).tolist() == [m.lookup(k) for k in ks]
    # (c) the tool front end and the Python mirror
    prefix = str(tmp_path / "a13")
    tf.tofile(prefix + ".tf.bin")
    rp = str(tmp_path / "synth.reads")
    open(rp, "wb").write(reads)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([os.path.join(root, "bin", "compute_aindex13"), rp, pf13_path(), prefix + ".tf.bin", prefix, "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.fromfile(prefix + ".index.bin", dtype=np.uint64), opos)
    assert np.array_equal(np.fromfile(prefix + ".indices.bin", dtype=np.uint64), oind)
    # the reference binary's own reading of the tf file (--tf-u32 / AIX_REF_COMPAT=1): its files, byte for byte, without test-side massaging
    z = np.load(os.path.join(gold, "count13", "expected.npz"))
    tf_ref = np.zeros(4 ** 13, dtype=np.uint64)
    tf_ref[z["synth.txt.idx"].astype(np.int64)] = z["synth.txt.cnt"]
    tf_ref.tofile(prefix + ".ref.tf.bin")                                           # what the reference's count_kmers13 wrote for these reads
    for how in ("flag", "env"):
        cmd = [os.path.join(root, "bin", "compute_aindex13"), rp, pf13_path(), prefix + ".ref.tf.bin", prefix + ".ref", "4"] + (["--tf-u32"] if how == "flag" else [])
        r = subprocess.run(cmd, env=dict(os.environ, **({"AIX_REF_COMPAT": "1"} if how == "env" else {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert np.array_equal(np.fromfile(prefix + ".ref.index.bin", dtype=np.uint64), g["positions"])
        assert hashlib.sha256(open(prefix + ".ref.indices.bin", "rb").read()).digest() == bytes(g["indices_sha256"])
        os.remove(prefix + ".ref.index.bin"); os.remove(prefix + ".ref.indices.bin")
    os.remove(prefix + ".ref.tf.bin")
    os.symlink(pf13_path(), prefix + ".pf")
    from aindex_amd.wrapper import AindexWrapper
    w = AindexWrapper()
    w.load_from_prefix_13mer(prefix)
    assert w.get_positions_13mer(reads[:13].decode()) == []                        # nothing mapped yet: the reference's permanent answer
    w.load_aindex_from_prefix_13mer(prefix)
    for k in (reads[:13], reads[150:163], b"ACGTACGTACGTA", b"AAAAAAAAAAAAA"):
        h = m.lookup(k)
        seg = opos[int(oind[h]):int(oind[h + 1])]
        assert w.get_positions_13mer(k.decode()) == (seg[seg != 0] - np.uint64(1)).tolist()
        assert w.get_positions(k.decode()) == w.get_positions_13mer(k.decode())
    assert w.get_positions_13mer("ACGTNACGTACGT") == [] and w.get_positions_13mer("acgtacgtacgta") == [] and w.get_positions_13mer("ACGT") == []
    ind_b, pos_b = w.build_aindex(rp)
    assert np.array_equal(pos_b, opos)
    w.load_aindex_from_prefix_13mer(prefix, ref_compat=True)                          # the reference's behaviour on request: positions never mapped
    assert w.get_positions_13mer(reads[:13].decode()) == []
    w.close()
    # a 13-mer table with an entry above 2^32 - 1 is refused (32-bit fill counters), never mis-filled
    with Index.open_13(pf13_path(), None) as ixb:
        big = tf.copy()
        big[int(np.nonzero(tf)[0][0])] = (1 << 32) + 3
        ixb.set_tf_13(big)
        with pytest.raises(_lib.AixError) as ei:
            ixb.positions_fill(reads[:5000])
        assert ei.value.status == -6
        ixb.set_tf_13(tf)
        assert np.array_equal(ixb.positions_fill(reads)[1], opos)


def test_kmer_counter_msd_path_equals_radix_path_and_oracle(monkeypatch):
    """K1 without a full-width sort (aix_k1.hip: two-level MSD partition + per-bucket LDS hash / sort) against the radix-sort
    path (AIX_K1_ROCPRIM=1) and the oracle: uniform reads, a heavy hitter (poly-A: one hash slot, many adds), k = 13 and k = 23,
    every canonical mode, min_count, and an input whose codes share their top 20 bits so that buckets overflow and the call falls
    back (same answer either way)."""
    from aindex_amd import counting
    rng = np.random.default_rng(5)
    asc = synth.genome_ascii(77, 400_000)
    reads = synth.reads_plain(43, asc, 6000, 150, rc_fraction_half=True, n_rate_ppm=2000).tobytes()
    polya = (b"A" * 150 + b"\n") * 3000 + (b"ACGT" * 37 + b"AC\n") * 500 + reads[:200_000]
    prefix = b"AAAAAAAAAA"
    narrow = b"".join(prefix + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 13)) + b"\n" for _ in range(60_000))
    # buckets of 500 .. 1 530 distinct keys (the in-LDS bitonic sort of a bucket, also past 1 024 keys; 1 536 is the most a bucket may
    # hold): 8-base prefixes pin the bucket (11 + 4 bits at this size), the rest is random
    pre8 = ["ACGTACGT", "CCGTTAGA", "GATTACAC", "TTGACCAG", "AGAGAGAT", "CTCTGGAA"]
    dense = b"".join(pre8[j].encode() + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 15)) + b"\n"
                     for j, m in enumerate((500, 800, 1100, 1300, 1450, 1530)) for _ in range(m))
    cases = [(reads, 23, 2, 1), (reads, 23, 1, 2), (reads, 13, 2, 1), (polya, 23, 1, 1), (polya, 23, 2, 3), (narrow, 23, 0, 1), (narrow, 13, 0, 1), (dense, 23, 0, 1), (b"", 23, 1, 1),
             (b"ACGTACGTACGTACGTACGTACG\n", 23, 1, 1)]
    for buf, k, mode, minc in cases:
        monkeypatch.delenv("AIX_K1_ROCPRIM", raising=False)
        keys, counts = counting.count_distinct(buf, k, mode, minc, fmt=_lib.FMT_PLAIN)
        monkeypatch.setenv("AIX_K1_ROCPRIM", "1")
        rkeys, rcounts = counting.count_distinct(buf, k, mode, minc, fmt=_lib.FMT_PLAIN)
        assert np.array_equal(keys, rkeys) and np.array_equal(counts, rcounts), (len(buf), k, mode, minc)
        okeys, ocnt = O.count_distinct(b">r\n" + buf.replace(b"\n", b"\n>r\n") if buf else buf, k, mode, minc)
        assert np.array_equal(keys, okeys) and np.array_equal(counts, ocnt.astype(np.uint64)), (len(buf), k, mode, minc)
    monkeypatch.delenv("AIX_K1_ROCPRIM", raising=False)
    monkeypatch.setenv("AIX_K1_TEST_REGION", "40")                 # an undersized chunk region must fail loudly, not drop k-mers
    with pytest.raises(_lib.AixError):
        counting.count_distinct(reads, 23, 2, 1, fmt=_lib.FMT_PLAIN)


def test_count23_histogram_backend_equals_atomics_and_oracle(canon_case, small23_prefix, monkeypatch):
    """k_count23_fixed's two back ends: one memory-side atomic per found window, or the slot stream + chunked-partition LDS
    histogram (no global atomics; the default for long buffers). Forced onto small inputs here, in one piece and cut into
    pieces at awkward places: same histogram as the atomics path and as the oracle, in every canonical mode; and the call
    ACCUMULATES into the caller's table (two calls = twice the counts)."""
    import torch
    orc = canon_case["orc"]
    asc = synth.genome_ascii(23, 300_000)
    reads = synth.reads_plain(41, asc, 4000, 150, rc_fraction_half=True, n_rate_ppm=1000).tobytes()
    noisy = bytes(asc[5000:9000]).lower() + b"\n" + bytes(asc[100:400]).replace(b"C", b"R", 2) + b"~U" + bytes(asc[900:1300]) + b"\n" + reads[:60_000]
    gold_reads = open(small23_prefix + ".reads", "rb").read()
    monkeypatch.setenv("AIX_MINIMIZER_TABLE", "1")                          # the minimizer-keyed copy + streaming probe kernel (experimental) are built
    prefix = canon_case["prefix"]
    with Index.open_23(small23_prefix + ".pf", small23_prefix + ".tf.bin", small23_prefix + ".kmers.bin") as gix:
        monkeypatch.setenv("AIX_MINIMIZER_LOAD", "14")                      # 14 keys per bucket on average: many buckets are longer than the 16 entries a lane reads, those windows go through the hash-keyed table
        cix = Index.open_23(prefix + ".pf", prefix + ".tf.bin", prefix + ".kmers.bin")
        monkeypatch.delenv("AIX_MINIMIZER_LOAD")
        monkeypatch.setenv("AIX_MINIMIZER_CAP", "3")                        # a lane reads three entries of a bucket: nearly every window is settled by the hash-keyed table
        tix = Index.open_23(prefix + ".pf", prefix + ".tf.bin", prefix + ".kmers.bin")
        monkeypatch.delenv("AIX_MINIMIZER_CAP")
        assert cix.info["minimizer_unfiled_keys"] > 0 and gix.info["minimizer_lines"] > 0 and tix.info["minimizer_unfiled_keys"] > cix.n // 4
        cases = [(cix, orc, reads), (cix, orc, noisy), (tix, orc, noisy), (gix, O.OracleIndex23.from_prefix(small23_prefix), gold_reads)]
        for ix, o, buf in cases:
            for mode in (0, 1, 2):
                want = o.count23_fixed(buf, False, mode)
                monkeypatch.setenv("AIX_COUNT23_ATOMICS", "1")
                assert np.array_equal(ix.count23_fixed(buf, _lib.FMT_PLAIN, mode), want)
                assert ix.info["count23_backend"] == 1
                monkeypatch.delenv("AIX_COUNT23_ATOMICS")
                monkeypatch.setenv("AIX_COUNT23_HIST_MIN", "0")
                for piece in (None, "1000", "77777"):
                    if piece is None:
                        monkeypatch.delenv("AIX_COUNT23_PIECE", raising=False)
                    else:
                        monkeypatch.setenv("AIX_COUNT23_PIECE", piece)
                    for bk, mk in ((True, True), (True, False), (False, False)):
                        ix.set_bucket_table(bk)
                        ix.set_minimizer_table(mk)
                        assert np.array_equal(ix.count23_fixed(buf, _lib.FMT_PLAIN, mode), want), (mode, piece, bk, mk)
                    # (True, False) with several pieces runs the probe of the next piece on the handle's second stream; the same on one stream
                    ix.set_bucket_table(True)
                    monkeypatch.setenv("AIX_COUNT23_OVERLAP", "0")
                    assert np.array_equal(ix.count23_fixed(buf, _lib.FMT_PLAIN, mode), want), (mode, piece, "one stream")
                    monkeypatch.delenv("AIX_COUNT23_OVERLAP")
                    # a key set beyond the 2^26 slots one pass of the histogram back end holds takes ceil(n / range) passes over the
                    # slot stream (range shrunk to 2^10 ... 2^14 slots here so that these small key sets need 2 ... 300 of them)
                    assert ix.info["count23_backend"] == 2 and ix.info["count23_passes"] == 1
                    for bits in ("10", "12", "14"):
                        monkeypatch.setenv("AIX_COUNT23_TEST_RANGE_BITS", bits)
                        assert np.array_equal(ix.count23_fixed(buf, _lib.FMT_PLAIN, mode), want), (mode, piece, "ranges", bits)
                        assert ix.info["count23_backend"] == 2 and ix.info["count23_passes"] == -(-ix.n // (1 << int(bits)))
                    monkeypatch.delenv("AIX_COUNT23_TEST_RANGE_BITS")
                    # the run-per-lane probe kernel (k_run23_slots: 16 / 32 consecutive windows per lane, bytes encoded once per run), every lane width
                    for run in ("16", "32"):
                        monkeypatch.setenv("AIX_COUNT23_RUN", run)
                        for lanes in (2, 8, 1, 4):
                            ix.set_bucket_table(True, lanes)
                            assert np.array_equal(ix.count23_fixed(buf, _lib.FMT_PLAIN, mode), want), (mode, piece, "run", run, lanes)
                    monkeypatch.delenv("AIX_COUNT23_RUN")
                    ix.set_bucket_table(True, 2)
                ix.set_bucket_table(True)
                # back end 3: the distinct k-mers of the buffer first (K1, in one piece and in pieces merged), one probe per distinct k-mer
                monkeypatch.setenv("AIX_COUNT23_VIA_K1", "1")
                for dpiece in (None, "5000", "33333"):
                    if dpiece is None:
                        monkeypatch.delenv("AIX_DISTINCT_PIECE", raising=False)
                    else:
                        monkeypatch.setenv("AIX_DISTINCT_PIECE", dpiece)
                    for bk in (True, False):
                        ix.set_bucket_table(bk)
                        assert np.array_equal(ix.count23_fixed(buf, _lib.FMT_PLAIN, mode), want), (mode, "via K1", dpiece, bk)
                        assert ix.info["count23_backend"] == 3
                ix.set_bucket_table(True)
                monkeypatch.delenv("AIX_DISTINCT_PIECE", raising=False)
                monkeypatch.delenv("AIX_COUNT23_VIA_K1")
                ix.set_minimizer_table(True)
                monkeypatch.delenv("AIX_COUNT23_PIECE", raising=False)
                t = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
                acc = torch.zeros(ix.n, dtype=torch.int32, device="cuda")
                ix.count23_fixed_t(t, mode, acc)
                ix.count23_fixed_t(t, mode, acc)
                torch.cuda.synchronize()
                assert np.array_equal(acc.cpu().numpy().view(np.uint32), 2 * want)
                monkeypatch.delenv("AIX_COUNT23_HIST_MIN")
        cix.close()
        tix.close()


def test_bucket_table_every_consumer_on_off_and_overflowing(canon_case, small23_prefix, monkeypatch):
    """Verification table (one 128-byte line per probe): every consumer — tf / total / both / kid+strand / 2-bit codes /
    coverage / count23 (three canonical modes) / positions fill — gives the answers of the MPHF path and of the oracle, for
    every lane width, on the canonical synthetic index, on the reference-built golden index (non-canonical keys: two probes)
    and on tables loaded so heavily (AIX_BUCKET_LOAD=8: ~40 % of the buckets overflow) that the fall-back runs all the time."""
    orc, g = canon_case["orc"], canon_case["genome"]
    prefix = canon_case["prefix"]
    q = mixed_queries(g, 120_000, 77)
    asc = synth.genome_ascii(23, 300_000)
    reads = synth.reads_plain(41, asc, 2500, 150, rc_fraction_half=True, n_rate_ppm=1000).tobytes()
    noisy = bytes(asc[5000:9000]).lower() + b"\n" + bytes(asc[100:400]).replace(b"C", b"R", 2) + b"~" + bytes(asc[900:1300]) + b"\n" + reads[:40_000]
    seqs = [bytes(asc[1000:9000]), bytes(asc[20000:20030]), b"ACGT", b"", bytes(asc[9000:12000]).lower(), bytes(asc[40000:41000]).replace(b"A", b"N", 3)]
    codes = synth.encode_kmers(synth.decode_kmers(synth.rolling_codes(g[:40_000], 23), 23))

    def answers(ix):
        kid, strand = ix.kid_strand_ascii(q)
        f, r = ix.both_ascii(q[:20_000])
        ind, pos = ix.positions_fill(noisy)
        return {"tf": ix.tf_ascii(q), "total": ix.total_ascii(q[:20_000]), "fwd": f, "rc": r, "kid": kid, "strand": strand, "codes": ix.tf_codes(codes),
                "cov": np.concatenate(ix.coverage(seqs, 0)), "c0": ix.count23_fixed(noisy, _lib.FMT_PLAIN, 0), "c1": ix.count23_fixed(noisy, _lib.FMT_PLAIN, 1),
                "c2": ix.count23_fixed(reads, _lib.FMT_PLAIN, 2), "ind": ind, "pos": pos}

    monkeypatch.setenv("AIX_MINIMIZER_TABLE", "1")                          # the experimental minimizer-keyed copy is built too
    for load in (None, "8", "0.5"):
        if load is None:
            monkeypatch.delenv("AIX_BUCKET_LOAD", raising=False)
            monkeypatch.delenv("AIX_MINIMIZER_LOAD", raising=False)
        else:
            monkeypatch.setenv("AIX_BUCKET_LOAD", load)
            monkeypatch.setenv("AIX_MINIMIZER_LOAD", load)
        for pre, o in ((prefix, orc), (small23_prefix, None)):
            with Index.open_23(pre + ".pf", pre + ".tf.bin", pre + ".kmers.bin") as ix:
                info = ix.info
                assert info["bucket_table"] == 1 and info["buckets"] >= 1
                if load == "8" and pre == prefix:
                    assert info["bucket_unfiled_keys"] > ix.n // 20          # the fall-back really is exercised
                ix.set_bucket_table(False)
                base = answers(ix)
                if o is not None:
                    assert np.array_equal(base["tf"], o.tf_batch(q, threads=8))
                    assert np.array_equal(base["c2"], o.count23_fixed(reads, False, 2))
                    oind, opos = o.positions(noisy)
                    assert np.array_equal(base["ind"], oind) and np.array_equal(base["pos"], opos)
                for lanes in (8, 4, 2, 1):
                    ix.set_bucket_table(True, lanes)
                    ix.set_absence_filter(lanes != 4)                        # the absence filter in front of the table: on, and off once
                    ix.set_minimizer_table(lanes != 2)                       # the minimizer-keyed copy for the streaming consumers: on, and off once
                    info = ix.info
                    assert info["bucket_lanes"] == lanes and (info["absence_filter_words"] > 0) == (lanes != 4) and (info["minimizer_lines"] > 0) == (lanes != 2)
                    got = answers(ix)
                    for k, v in base.items():
                        assert np.array_equal(got[k], v), (load, pre, lanes, k)

    monkeypatch.setenv("AIX_BUCKET_TABLE", "0")                              # not built at all: the MPHF path alone
    with Index.open_23(prefix + ".pf", prefix + ".tf.bin", prefix + ".kmers.bin") as ix:
        assert ix.info["bucket_table"] == 0 and ix.info["buckets"] == 0
        ix.set_bucket_table(True, 8)                                         # nothing to switch on
        assert np.array_equal(ix.tf_ascii(q), orc.tf_batch(q, threads=8))


# ------------------------------------------------------------------------------------------------
# N > 1 on the GPU: every sharded entry point of aindex_amd.dist (tests/_dist_gpu_worker.py). Two gloo ranks share
# cuda:0 (this box has one GPU); one nccl rank runs the same code through RCCL (all_reduce, all_to_all, barrier).
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend,nproc", [("gloo", 2), ("nccl", 1)])
def test_sharded_entry_points_multi_process(backend, nproc):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", AIX_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nproc == 1:
        env["AIX_FORCE_DIST"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", "29683" if nproc == 2 else "29684", os.path.join(root, "tests", "_dist_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and f"DIST_GPU_OK {backend} {nproc}" in out, out[-4000:]


def test_failing_host_batch_drains_its_streams(small23_prefix, tmp_path):
    """a large host batch that fails in the middle (AIX_PIPE_TEST_FAIL_CHUNK: the launch of chunk 2 "fails") returns with its
    three pipe streams drained, so the next call on the handle — which reuses the same pinned and device staging — answers correctly.
    The hook is read once per process, hence the child process."""
    import subprocess
    import sys
    code = f"""
import sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, "tests")!r})
import numpy as np
import oracle_lib as O
from aindex_amd import _lib, synth
from aindex_amd.engine import Index
prefix = {small23_prefix!r}
orc = O.OracleIndex23.from_prefix(prefix)
keys = synth.decode_kmers(orc.checker(), 23)
q = np.ascontiguousarray(np.tile(keys, (2200, 1)))[:9_000_000].reshape(-1)          # 9 M queries = 5 chunks of 2 M
ix = Index.open_23(prefix + ".pf", prefix + ".tf.bin", prefix + ".kmers.bin")
try:
    ix.tf_ascii(q)
    print("NO_ERROR")
except _lib.AixError as e:
    print("FAILED_AS_ASKED", e.status)
small = q[: 3_000_000 * 23]                                                            # 2 chunks: never reaches the failing chunk
got = ix.tf_ascii(small)
print("SECOND_CALL_OK" if np.array_equal(got, orc.tf_batch(small, threads=8)) else "SECOND_CALL_WRONG")
"""
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AIX_PIPE_TEST_FAIL_CHUNK="2"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "FAILED_AS_ASKED -5" in out and "SECOND_CALL_OK" in out, out + r.stderr.decode()[-1500:]
