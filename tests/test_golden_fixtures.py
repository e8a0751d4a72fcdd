"""Committed regression fixtures (tests/golden/*.npz, made by make_golden.py from
this repo's oracle -- see the provenance note there).  CPU: the oracle and the
kernels-on-the-CPU-model must still reproduce them.  GPU: the HIP path must."""
from pathlib import Path

import numpy as np
import pytest

import jxlt_testlib as T

GOLDEN = sorted((Path(__file__).resolve().parent / "golden").glob("*.npz"))


def _check(fx, r, with_floats):
    assert np.array_equal(r.quant_dc, fx["quant_dc"])
    assert np.array_equal(r.raw_quant, fx["raw_quant"])
    assert np.array_equal(r.strategy, fx["strategy"])
    assert np.array_equal(r.ytox, fx["ytox"]) and np.array_equal(r.ytob, fx["ytob"])
    assert r.all_tokens() == fx["tokens"].tobytes()
    assert [len(t) for t in r.group_tokens] == list(fx["group_token_bytes"])
    if with_floats:
        assert np.array_equal(r.xyb.view(np.uint32), fx["xyb"].view(np.uint32))
        assert np.array_equal(r.qf.view(np.uint32), fx["quant_field"].view(np.uint32))
        assert np.array_equal(r.mask.view(np.uint32), fx["masking"].view(np.uint32))


@pytest.mark.parametrize("path", GOLDEN, ids=[p.stem for p in GOLDEN])
def test_oracle_reproduces_golden(built, path):
    fx = np.load(path)
    r = T.oracle_hot_path(fx["planes"], float(fx["distance"]), bool(fx["force_dct8"]))
    _check(fx, r, True)
    if not bool(fx["force_dct8"]):
        with T.reference_single_symbol_codes():  # the fixtures hold the reference's bytes
            assert T.assemble_codestream(r, float(fx["distance"])) == fx["codestream"].tobytes()


@pytest.mark.parametrize("path", GOLDEN[:3], ids=[p.stem for p in GOLDEN[:3]])
def test_kernels_on_cpu_model_reproduce_golden(built, path):
    fx = np.load(path)
    r = T.sim_hot_path(fx["planes"], float(fx["distance"]), bool(fx["force_dct8"]))
    _check(fx, r, True)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[p.stem for p in GOLDEN])
def test_gpu_reproduces_golden(built, path):
    fx = np.load(path)
    enc = built.Encoder(0)
    r = enc.hot_path(fx["planes"], float(fx["distance"]), force_dct8=bool(fx["force_dct8"]), debug=True)
    _check(fx, r, True)
    if not bool(fx["force_dct8"]):
        with T.reference_single_symbol_codes():
            assert built.encode_file(fx["planes"], float(fx["distance"])) == fx["codestream"].tobytes()
    enc.close()
