"""bench.py's host-side logic that can be checked without a GPU: the attribution of slow steps."""
import sys
import time

import jxlt_testlib as T

sys.path.insert(0, str(T.ROOT))


def test_slow_steps_get_a_cause():
    import bench
    probe = bench.StepProbe()
    probe.sample()
    for _ in range(5):
        time.sleep(0.001)
        probe.sample()
    usual = {"tile_kernel": 4.0, "tokenisation_after_tile_kernel": 0.55}
    kernels = [usual, usual, {"tile_kernel": 4.6, "tokenisation_after_tile_kernel": 0.6}, dict(usual, tile_kernel=4.2), usual]
    copies = [(4, 20.0), (4, 20.0), (4, 2300.0), (4, 21.0), (4, 21.0)]
    d = bench.step_diagnostics([5.2, 5.21, 8.1, 5.5, 5.19], kernels, copies, probe)
    assert d["median_ms"] == 5.21 and d["mean_minus_median_ms"] > 0.5
    assert [s["step"] for s in d["slow_steps"]] == [2]
    cause = d["slow_steps"][0]["cause"]
    assert cause["copy_call_on_host_ms"] > 2.0 and cause["device_kernels_ms"] > 0.5
    assert d["totals"]["longest_copy_call_us"] == 2300.0
    # a slow step with nothing to show for it says so
    flat = bench.step_diagnostics([5.2, 5.2, 7.0, 5.2], [usual] * 4, [(4, 20.0)] * 4, None)
    assert flat["slow_steps"][0]["cause"] == {"unattributed_ms": 1.8}
    # and a flat run lists nothing
    assert bench.step_diagnostics([5.2, 5.21, 5.19], [usual] * 3, [(4, 20.0)] * 3, None)["slow_steps"] == []


def test_a_counter_profile_of_other_kernel_sources_is_called_stale(capsys):
    """roofline.traffic / valu_issue come from committed counter profiles; the line must say loudly
    when those were collected with other device sources than the ones that run."""
    import bench
    now = bench.kernel_source_sha16()
    fresh = bench.profile_freshness("profiles/x_traffic_1.json", {"kernel_source_sha16": now})
    assert fresh["stale"] is False and fresh["stale_reason"] is None
    assert capsys.readouterr().err == ""
    old = bench.profile_freshness("profiles/x_traffic_1.json", {"kernel_source_sha16": "0123456789abcdef"})
    assert old["stale"] is True and "have changed since" in old["stale_reason"]
    assert "WARNING" in capsys.readouterr().err
    none = bench.profile_freshness("profiles/x_traffic_1.json", {})
    assert none["stale"] is True and "no fingerprint" in none["stale_reason"]
