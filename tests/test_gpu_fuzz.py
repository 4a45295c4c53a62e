"""Differential fuzzing: many small random indexes / query sets / read buffers, HIP path vs the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_lib as O
from aindex_amd import _lib, builder, synth
from aindex_amd.engine import Index

ALPH = np.frombuffer(b"ACGTACGTACGTNacgtn~\n?U", dtype=np.uint8)


def _seeds(n):
    """The suite runs seeds 0..n-1; AIX_FUZZ_SEEDS="lo:hi" swaps in another range for one-off soak runs on a GPU box."""
    r = os.environ.get("AIX_FUZZ_SEEDS")
    if r:
        lo, hi = r.split(":")
        return range(int(lo), int(hi))
    return range(n)


def make_case(seed, tmp):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 3, 4, 7, 50, 333, 2000, 6000]))
    codes = np.unique(rng.integers(0, 4 ** 23, size=n, dtype=np.uint64))
    mode = seed % 3
    if mode == 0:                                  # all true-canonical -> fast path
        codes = np.unique(np.minimum(codes, synth.revcomp_codes(codes, 23)))
    elif mode == 1:                                # both strands of some keys stored
        codes = np.unique(np.concatenate([codes, synth.revcomp_codes(codes[: max(1, len(codes) // 3)], 23)]))
    if codes.shape[0] == 2:                        # hash domain 1 is never peelable (also in the reference)
        codes = codes[:1]
    tfs = rng.integers(0, 1000, size=codes.shape[0]).astype(np.uint32)
    tfs[rng.integers(0, codes.shape[0])] = 0xFFFFFFFF
    pf = builder.build_pf_codes(codes, 23)
    prefix = os.path.join(tmp, f"f{seed}")
    open(prefix + ".pf", "wb").write(pf)
    m = O.OracleMphf(prefix + ".pf")
    keys = synth.decode_kmers(codes, 23)
    rc, checker, tf = O.index_scatter(m, np.ascontiguousarray(keys).reshape(-1), tfs)
    assert rc == 0
    checker.tofile(prefix + ".kmers.bin")
    tf.tofile(prefix + ".tf.bin")
    return rng, codes, keys, prefix


@pytest.mark.parametrize("seed", _seeds(24))
def test_fuzz_queries_counts_positions(seed, tmp_path):
    rng, codes, keys, prefix = make_case(seed, str(tmp_path))
    orc = O.OracleIndex23.from_prefix(prefix)
    with Index.open_23(prefix + ".pf", prefix + ".tf.bin", prefix + ".kmers.bin") as ix:
        assert ix.canonical_only == bool(np.all(codes <= synth.revcomp_codes(codes, 23)))
        nq = 3000
        q = ALPH[rng.integers(0, ALPH.shape[0], size=(nq, 23))].copy()
        pick = rng.integers(0, keys.shape[0], size=nq // 3)
        q[: nq // 3] = keys[pick]
        q[nq // 3: nq // 2] = synth.decode_kmers(synth.revcomp_codes(codes[pick[: nq // 2 - nq // 3]], 23), 23)
        mut = rng.integers(0, nq // 2, size=200)
        q[mut, rng.integers(0, 23, size=200)] = ALPH[rng.integers(0, ALPH.shape[0], size=200)]
        want_tf = orc.tf_batch(q)
        qb = [bytes(x) for x in q[:400]]
        for fast in (True, False):
            for ee in (True, False):
                for bk, lanes in ((True, 8), (True, 1 << (seed % 3)), (False, 0)):     # verification table on (two lane widths) / off
                    ix.set_canonical_fastpath(fast); ix.set_early_exit(ee); ix.set_bucket_table(bk, lanes)
                    assert np.array_equal(ix.tf_ascii(q), want_tf), (seed, fast, ee, bk, lanes)
        ix.set_canonical_fastpath(True); ix.set_early_exit(True); ix.set_bucket_table(seed % 2 == 0, 8)
        assert ix.total_ascii(q[:400]).tolist() == [orc.total(b) for b in qb]
        f, r = ix.both_ascii(q[:400])
        assert [(int(a), int(b)) for a, b in zip(f, r)] == [orc.both(b) for b in qb]
        kid, strand = ix.kid_strand_ascii(q[:400])
        assert strand.tolist() == [orc.strand(b) for b in qb] and kid.tolist() == [orc.kid(b) for b in qb]
        assert np.array_equal(ix.hash_ascii(q[:400]), orc.hash_batch(q[:400]))
        # ragged lengths
        items = [bytes(x)[: int(rng.integers(0, 24))] + bytes(ALPH[rng.integers(0, 8, size=int(rng.integers(0, 30)))]) for x in q[:300]]
        assert ix.tf_ragged(items).tolist() == [orc.tf(b) for b in items]
        # a reads-like buffer built from keys, their reverse complements and noise, with separators
        parts = []
        for i in range(60):
            k1 = bytes(keys[int(rng.integers(0, keys.shape[0]))])
            noise = bytes(ALPH[rng.integers(0, ALPH.shape[0], size=int(rng.integers(0, 40)))])
            parts.append(k1 + noise + bytes(q[int(rng.integers(0, nq))]) + (b"\n" if i % 3 else b"~"))
        buf = b"".join(parts)
        for mode in (0, 1, 2):
            assert np.array_equal(ix.count23_fixed(buf, _lib.FMT_PLAIN, mode), orc.count23_fixed(buf, False, mode)), (seed, mode)
        ind, pos = ix.positions_fill(buf)
        oind, opos = orc.positions(buf)
        assert np.array_equal(ind, oind) and np.array_equal(pos, opos)
        # the multi-GPU shard protocols, replayed rank by rank: positions (tallies + carried slot numbering) and index scatter
        from shard_helpers import positions_by_shards, scatter_by_shards
        world = 2 + seed % 3
        pfa = np.fromfile(prefix + ".pf", dtype=np.uint8)
        ck, tfv = np.fromfile(prefix + ".kmers.bin", dtype=np.uint64), np.fromfile(prefix + ".tf.bin", dtype=np.uint32)
        # (one tf of the case is 2^32-1, i.e. a 34 GB positions array per call: the shard replay runs on the same keys with tf <= 3)
        with Index.create_23(pfa.tobytes(), ck, np.minimum(tfv, 3).astype(np.uint32)) as ix_small:
            wind, wpos = ix_small.positions_fill(buf)
            sind, spos, _ = positions_by_shards(ix_small, buf, world)
            assert np.array_equal(sind, wind) and np.array_equal(spos, wpos), (seed, world)
        n = ck.shape[0]
        perm = rng.permutation(n)                                         # the .dat order is not the slot order
        cuts = [0] + sorted(int(x) for x in rng.integers(0, n + 1, size=world - 1)) + [n]
        sc, stf, socc, sts, clash = scatter_by_shards(pfa, synth.decode_kmers(ck, 23)[perm], tfv[perm], n, cuts)
        assert all(x == 0 for x in sts) and not clash, (seed, sts)
        assert np.array_equal(sc, ck) and np.array_equal(stf, tfv)
        seqs = [buf[:200], buf[200:460], b"", buf[-30:]]
        for cutoff in (0, 5):
            for s, got in zip(seqs, ix.coverage(seqs, cutoff)):
                assert np.array_equal(got, orc.coverage(s, cutoff))


# ------------------------------------------------------------------------------------------------
# 13-mer mode: random buffers in the three input formats, then noisy queries against the counted table
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ix13():
    from pf13 import pf13_path
    ix = Index.open_13(pf13_path(), None)
    yield ix
    ix.close()


def _rand_seq(rng, lo, hi):
    return bytes(ALPH[rng.integers(0, ALPH.shape[0] - 3, size=int(rng.integers(lo, hi)))])     # no '~', '\n', '?' inside


@pytest.mark.parametrize("seed", _seeds(8))
def test_fuzz_13mer(seed, ix13):
    from pf13 import pf13_path
    rng = np.random.default_rng(1_000_000 + seed)
    m = O.OracleMphf(pf13_path())
    kind = seed % 4
    recs = [_rand_seq(rng, 0, 120) for _ in range(int(rng.integers(1, 60)))]
    if kind == 0:
        buf = b"".join(r + (b"\n" if rng.random() < 0.9 else b"~" + _rand_seq(rng, 0, 50) + b"\n") for r in recs)
    elif kind == 1:
        buf = b"".join(b">h%d %s\n" % (i, _rand_seq(rng, 0, 10)) + b"".join(r[j:j + 37] + (b"\r\n" if i % 7 == 0 else b"\n") for j in range(0, len(r), 37)) +
                       (b"\n" if i % 5 == 0 else b"") for i, r in enumerate(recs))
    elif kind == 2:
        buf = b"".join(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(recs))
    else:
        buf = b"\n" + b"".join(r + b"\n" for r in recs)                    # empty first line -> PLAIN
    if seed >= 4:
        buf = buf.rstrip(b"\n")                                            # no trailing newline
    want = O.count13(m, buf, -1)
    got = ix13.count13(buf)
    assert np.array_equal(got, want), (seed, kind)
    tf = want.copy()
    tf[np.nonzero(tf)[0][::3]] += np.uint64(1 << 33)                       # > 32-bit values
    ix13.set_tf_13(tf)
    orc = O.OracleIndex13(pf13_path(), tf)
    plain = [r for r in recs if len(r) >= 13]
    q = []
    for r in plain[:40]:
        p = int(rng.integers(0, len(r) - 12))
        q.append(r[p:p + 13])
    q += [bytes(ALPH[rng.integers(0, ALPH.shape[0], size=13)]) for _ in range(200)]
    q += [bytes(ALPH[rng.integers(0, 12, size=13)]) for _ in range(200)]   # mostly valid upper-case
    qa = np.frombuffer(b"".join(q), dtype=np.uint8)
    assert np.array_equal(ix13.tf_ascii(qa), orc.tf_batch(qa))
    assert ix13.total_ascii(qa).tolist() == [orc.total(s) for s in q]
    f, r2 = ix13.both_ascii(qa)
    assert [(int(a), int(b)) for a, b in zip(f, r2)] == [orc.both(s) for s in q]
    items = [s[: int(rng.integers(0, 14))] + bytes(ALPH[rng.integers(0, 8, size=int(rng.integers(0, 5)))]) for s in q[:100]]
    assert ix13.tf_ragged(items).tolist() == [orc.tf(s) for s in items]
    for s, got_cov in zip(plain[:5], ix13.coverage(plain[:5], 1)):
        assert np.array_equal(got_cov, orc.coverage(s, 1))


@pytest.mark.parametrize("seed", _seeds(6))
def test_fuzz_normalise_and_distinct(seed):
    """Random FASTA/FASTQ-like byte soup: device normalisation == host normalisation; distinct k-mer sets == oracle."""
    import torch
    from aindex_amd import counting
    rng = np.random.default_rng(2_000_000 + seed)
    soup = np.frombuffer(b"ACGTACGTACGTNacgtu>@+\n\n\r ~U", dtype=np.uint8)
    buf = bytes(soup[rng.integers(0, soup.shape[0], size=int(rng.integers(1, 40000)))])
    if seed % 2 == 0:
        buf = b">" + buf
    t = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    for fmt, mode in ((1, 0), (1, 1), (2, 0)):
        want = counting.normalize(buf, fmt, mode)
        got = counting.normalize_t(t, fmt, mode).cpu().numpy().tobytes()
        assert got == want, (seed, fmt, mode)
    for k in (5, 13, 17, 23, 31):
        for canon in (0, 1, 2):
            keys, counts = counting.count_distinct(buf, k, canon, 1 + seed % 2)
            okeys, ocnt = O.count_distinct(buf, k, canon, 1 + seed % 2)
            assert np.array_equal(keys, okeys) and np.array_equal(counts, ocnt), (seed, k, canon)
    # counted piece by piece (what buffers beyond 2^31 windows go through): every cut position gives the same set
    for piece in (1, 97, 4096):
        os.environ["AIX_DISTINCT_PIECE"] = str(piece)
        try:
            small = buf[: 3000 if piece == 1 else len(buf)]
            for k, canon, mc in ((13, 2, 1), (23, 1, 2), (31, 0, 1)):
                keys, counts = counting.count_distinct(small, k, canon, mc)
                okeys, ocnt = O.count_distinct(small, k, canon, mc)
                assert np.array_equal(keys, okeys) and np.array_equal(counts, ocnt), (seed, piece, k, canon)
        finally:
            del os.environ["AIX_DISTINCT_PIECE"]


@pytest.mark.parametrize("seed", _seeds(10))
def test_fuzz_corrupt_index_files(seed, tmp_path):
    """Index files that disagree with the MPHF (swapped / foreign / duplicated / out-of-range codes, short tf file,
    extra trailing entries): the HIP path must answer exactly like the reference's evaluator on the same files."""
    rng, codes, keys, prefix = make_case(3_000_000 + seed, str(tmp_path))
    checker = np.fromfile(prefix + ".kmers.bin", dtype=np.uint64)
    tf = np.fromfile(prefix + ".tf.bin", dtype=np.uint32)
    n = checker.shape[0]
    for _ in range(max(1, n // 10)):
        i, j = rng.integers(0, n, size=2)
        op = int(rng.integers(0, 5))
        if op == 0:
            checker[[i, j]] = checker[[j, i]]
        elif op == 1:
            checker[i] = rng.integers(0, 4 ** 23, dtype=np.uint64)
        elif op == 2:
            checker[i] = checker[j]
        elif op == 3:
            checker[i] |= np.uint64(1) << np.uint64(int(rng.integers(46, 64)))
        else:
            checker[i] = synth.revcomp_codes(checker[j:j + 1] & np.uint64(4 ** 23 - 1), 23)[0]
    extra = int(rng.integers(0, 4))
    if extra:
        checker = np.concatenate([checker, rng.integers(0, 4 ** 23, size=extra, dtype=np.uint64)])
        tf = np.concatenate([tf, rng.integers(1, 9, size=extra).astype(np.uint32)])
    p2 = os.path.join(str(tmp_path), "c")
    checker.tofile(p2 + ".kmers.bin")
    tf[: max(0, tf.shape[0] - int(rng.integers(0, 3)))].tofile(p2 + ".tf.bin")          # tf file may be short (hash.cpp:431-444)
    import shutil
    shutil.copy(prefix + ".pf", p2 + ".pf")
    orc = O.OracleIndex23.from_prefix(p2)
    allc = np.unique(np.concatenate([codes, checker & np.uint64(4 ** 23 - 1)]))
    q = np.concatenate([synth.decode_kmers(allc, 23), synth.decode_kmers(synth.revcomp_codes(allc, 23), 23),
                        ALPH[rng.integers(0, ALPH.shape[0], size=(500, 23))]])
    if q.shape[0] > 6000:
        q = q[rng.permutation(q.shape[0])[:6000]]
    q = q.copy()
    mut = rng.integers(0, q.shape[0], size=300)
    q[mut, rng.integers(0, 23, size=300)] = ALPH[rng.integers(0, ALPH.shape[0], size=300)]
    want = orc.tf_batch(q)
    qb = [bytes(x) for x in q[:300]]
    with Index.open_23(p2 + ".pf", p2 + ".tf.bin", p2 + ".kmers.bin") as ix:
        for fast in (True, False):
            for fp in (True, False):
                for ee in (True, False):
                    for bk in (True, False):
                        ix.set_canonical_fastpath(fast); ix.set_fingerprint_filter(fp); ix.set_early_exit(ee); ix.set_bucket_table(bk, 8 >> (seed % 4))
                        assert np.array_equal(ix.tf_ascii(q), want), (seed, fast, fp, ee, bk)
        ix.set_canonical_fastpath(True); ix.set_fingerprint_filter(True); ix.set_early_exit(True); ix.set_bucket_table(seed % 2 == 1, 8)
        kid, strand = ix.kid_strand_ascii(q[:300])
        assert strand.tolist() == [orc.strand(b) for b in qb] and kid.tolist() == [orc.kid(b) for b in qb]
        assert ix.total_ascii(q[:300]).tolist() == [orc.total(b) for b in qb]
        buf = b"\n".join(bytes(x) for x in q[:200]) + b"\n"
        for mode in (0, 1, 2):
            assert np.array_equal(ix.count23_fixed(buf, _lib.FMT_PLAIN, mode), orc.count23_fixed(buf, False, mode))
        ind, pos = ix.positions_fill(buf)
        oind, opos = orc.positions(buf)
        assert np.array_equal(ind, oind) and np.array_equal(pos, opos)
