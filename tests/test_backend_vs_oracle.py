"""The product's host bitstream back-end (libjxl-tiny_amd/host/: DC tokeniser, clustering,
Huffman construction, code serialisation, headers, TOC, assembly) against the ORACLE's
independent restatement of the same reference code (oracle/jxl_tiny_bitstream_oracle.c).

Every other test that compares a codestream compares it with the oracle's bytes; this file is
where the two back-ends meet, stage by stage and end to end, in both single-symbol modes."""
import numpy as np
import pytest

import jxlt_testlib as T

CASES = [
    # w, h, distance, hard
    (9, 7, 1.0, False), (17, 300, 1.0, False), (200, 137, 2.0, False), (256, 256, 1.0, False),
    (264, 260, 8.0, False), (300, 264, 0.3, False), (72, 40, 0.5, True), (520, 300, 4.0, False),
    (2100, 300, 1.0, False), (300, 2100, 16.0, False), (2060, 2060, 3.0, False), (512, 512, 1.0, True),
    (64, 64, 25.0, False),
]


@pytest.fixture(scope="module")
def results(built):
    out = {}
    for w, h, d, hard in CASES:
        out[(w, h, d, hard)] = T.oracle_hot_path(T.to_planes(T.synthetic_image(w, h, hard=hard)), d)
    return out


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_d%g%s" % (c[0], c[1], c[2], "_noise" if c[3] else ""))
def test_host_backend_codestream_equals_oracle_backend(built, results, case):
    res, d = results[case], case[2]
    assert T.host_assemble_codestream(res, d) == T.oracle_codestream(res, d, reference_single_symbol=False)
    with T.reference_single_symbol_codes():
        assert T.host_assemble_codestream(res, d) == T.oracle_codestream(res, d, reference_single_symbol=True)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_d%g%s" % (c[0], c[1], c[2], "_noise" if c[3] else ""))
def test_host_dc_tokeniser_equals_oracle_dc_tokeniser(built, results, case):
    res = results[case]
    assert T.host_dc_records(res) == T.oracle_dc_records(res)


def _histograms(res):
    ac = T.token_histogram(res.all_tokens())
    dc = sum((T.token_histogram(r) for r in T.oracle_dc_records(res)))
    return ac, dc


@pytest.mark.parametrize("case", CASES[:9], ids=lambda c: "%dx%d_d%g%s" % (c[0], c[1], c[2], "_noise" if c[3] else ""))
def test_code_tables_from_real_histograms(built, results, case):
    ac, dc = _histograms(results[case])
    for ref in (False, True):
        want = T.oracle_code_tables(ac, dc, reference_single_symbol=ref)
        if ref:
            with T.reference_single_symbol_codes():
                got = built.build_code_tables(ac, dc)
        else:
            got = built.build_code_tables(ac, dc)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.parametrize("seed", range(12))
def test_code_tables_from_random_histograms(built, seed):
    """Clustering and Huffman tie-breaks on adversarial statistics: empty contexts, single-symbol contexts,
    equal counts (ties everywhere), geometric counts deep enough for the 15-bit length limit to bite.
    (Totals stay below 2^32 like any real token count: beyond that the reference's own uint32 node sums wrap.)"""
    rng = np.random.default_rng(seed)

    def make(nctx):
        h = np.zeros((64, 64), np.uint32)
        for c in range(nctx):
            kind = rng.integers(0, 6)
            if kind == 0:
                continue
            n = int(rng.integers(1, 64))
            sym = rng.choice(64, size=n, replace=False)
            if kind == 1:
                h[c, sym[0]] = rng.integers(1, 1000)
            elif kind == 2:
                h[c, sym] = 7
            elif kind == 3:
                h[c, sym] = (1 << np.minimum(np.arange(n), 20)).astype(np.uint32)
            else:
                h[c, sym] = rng.integers(1, 1 << int(rng.integers(2, 20)), size=n)
        if not h.any():
            h[0, 0] = 1
        return h

    ac, dc = make(64), make(45)
    want = T.oracle_code_tables(ac, dc, reference_single_symbol=False)
    got = built.build_code_tables(ac, dc)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.parametrize("seed", range(6))
def test_code_tables_when_many_histograms_are_alike(built, seed):
    """The selection rounds of the clustering are evaluated lazily since round 4 (a histogram whose upper bound is
    below another's exact distance is not compared with the newest cluster at all): the choices must be the
    reference's also where distances TIE -- copies of one histogram, copies scaled by a constant, near copies that
    differ in one count, zero distances, fewer distinct histograms than clusters."""
    rng = np.random.default_rng(1000 + seed)

    def make(nctx):
        h = np.zeros((64, 64), np.uint32)
        nbase = int(rng.integers(1, 12))
        base = []
        for _ in range(nbase):
            n = int(rng.integers(1, 40))
            b = np.zeros(64, np.uint32)
            b[rng.choice(64, size=n, replace=False)] = rng.integers(1, 1 << int(rng.integers(1, 16)), size=n)
            base.append(b)
        for c in range(nctx):
            kind = int(rng.integers(0, 5))
            b = base[int(rng.integers(0, nbase))].copy()
            if kind == 0:
                continue                      # empty context
            if kind == 2:
                b = b * np.uint32(rng.integers(2, 5))   # the same shape, another weight
            if kind == 3:
                b[int(rng.integers(0, 64))] += 1        # a near copy
            h[c] = b
        if not h.any():
            h[0, 0] = 1
        return h

    for _ in range(10):
        ac, dc = make(64), make(45)
        want = T.oracle_code_tables(ac, dc, reference_single_symbol=False)
        got = built.build_code_tables(ac, dc)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.parametrize("distance", ["0.1", "0.5", "1", "4"])
def test_code_tables_from_the_bench_frames_histograms(built, distance):
    """The histograms of the 16384 x 16384 bench frame at four distances (tests/golden/histograms/bench_histograms.npz: counts
    up to 10^8, written on the GPU box by tools/dump_histograms.py -- data, not code).  Four in ten of the
    clustering's Huffman costs on the DC histograms end in a tree deeper than 15 bits and take the depth-limited
    rounds of host/entropy_coder.cc (HuffmanBitCost); the tables must be the oracle's all the same."""
    h = np.load(T.ROOT / "tests" / "golden" / "histograms" / "bench_histograms.npz")
    ac, dc = h["ac_" + distance], h["dc_" + distance]
    want = T.oracle_code_tables(ac, dc, reference_single_symbol=False)
    got = built.build_code_tables(ac, dc)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_sections_packed_with_oracle_tables_assemble_to_oracle_codestream(built, results):
    """The production route (code tables -> packed sections -> jxlt_finish_frame: header, TOC, globals by
    the host library) against the oracle's codestream, with the sections packed by the test's own
    reference packer from the ORACLE's tables."""
    case = (2100, 300, 1.0, False)
    res = results[case]
    ac, dc = _histograms(res)
    at, dt = T.oracle_code_tables(ac, dc)

    def pk(sections, table):
        packed = T.pack_sections_python(sections, table)
        off = np.zeros(len(packed) + 1, np.uint64)
        off[1:] = np.cumsum([len(p[0]) for p in packed])
        return np.frombuffer(b"".join(p[0] for p in packed), np.uint8), off, np.array([p[1] for p in packed], np.uint32)

    got = built.finish_frame(case[0], case[1], case[2], ac, dc, pk(T.oracle_dc_records(res), dt), pk(res.group_tokens, at))
    assert got == T.oracle_codestream(res, case[2])
