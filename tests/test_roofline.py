"""bench.py's roofline object can only quote rocprofv3 counters of the build it is running: profiles carry
the git blob hashes of the kernel sources, and a mismatch turns achieved / frac into null plus the reason."""
import copy
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stamp_is_git_hash_object_and_staleness_is_detected(pkg):
    from rtiow_rust_amd import roofline as rl
    st = rl.source_stamp(ROOT)
    rel = "rtiow-rust_amd/csrc/rt_pool.h"
    if os.path.isdir(os.path.join(ROOT, ".git")):
        assert st["blobs"][rel] == subprocess.check_output(["git", "hash-object", os.path.join(ROOT, rel)]).decode().strip()
    pmc = {"counters_avg_per_launch": {"SQ_INSTS_VALU": 1e9, "SQ_THREAD_CYCLES_VALU": 32e9, "SQ_ACTIVE_INST_VALU": 1e9},
           "samples_per_launch": 100, "build": copy.deepcopy(st)}
    assert rl.profile_staleness(pmc, ROOT) is None
    fresh = rl.valu_roofline(pmc, 1e-3, samples=100, stale=rl.profile_staleness(pmc, ROOT))
    assert fresh["frac"] is not None and 0 < fresh["frac"] < 1 and "stale_profile" not in fresh
    # another build of the library, other schedule options, changed sources, an unstamped (round-2) profile
    assert "RTIOW_GPU_LIB" in rl.profile_staleness(pmc, ROOT, lib_override="/tmp/other.so")
    assert "RTG_REFILL_MIN" in rl.profile_staleness(pmc, ROOT, knobs=["RTG_REFILL_MIN"])
    pmc["build"]["blobs"][rel] = "0" * 40
    pmc["build"]["digest"] = "1" * 40
    why = rl.profile_staleness(pmc, ROOT)
    assert why and "rt_pool.h" in why
    stale = rl.valu_roofline(pmc, 1e-3, samples=100, stale=why)
    assert stale["frac"] is None and stale["achieved"] is None and stale["stale_profile"] == why and stale["peak"] > 0
    del pmc["build"]
    assert "no build stamp" in rl.profile_staleness(pmc, ROOT)


def test_current_profiles_name_existing_files(pkg):
    import json
    cur = json.load(open(os.path.join(ROOT, "profiles", "current.json")))
    for k, rel in cur.items():
        assert os.path.exists(os.path.join(ROOT, rel)), (k, rel)


def test_current_profiles_belong_to_this_build(pkg):
    """The counters profiles/current.json points at were collected on THIS tree's kernel sources (else bench.py reports
    roofline.frac = null with the reason).  A kernel edit makes them stale until tools/collect_profiles.sh has run again on
    the GPU box: that is reported as a skip here, not as a failure -- the round's last commit has to pass it."""
    import json
    import pytest
    from rtiow_rust_amd import roofline as rl
    cur = json.load(open(os.path.join(ROOT, "profiles", "current.json")))
    stale = {}
    for k, rel in cur.items():
        why = rl.profile_staleness(rl.load_pmc(os.path.join(ROOT, rel)), ROOT)
        if why:
            stale[k] = why
    if stale:
        pytest.skip("stale counter profiles (re-run tools/collect_profiles.sh): %s" % stale)


def test_algorithmic_valu_prices_reference_work_only(pkg):
    """roofline.algorithmic_valu: c_box N + c_prim P + c_shade H + c_cam samples over the f32 lane peak; costs of
    another build are refused like counters of another build."""
    from rtiow_rust_amd import roofline as rl
    costs = {"kernels": {"render_lean_pool": {"kernel": "k", "c_box": 20.0, "c_prim": 50.0, "c_shade": 300.0, "c_cam": 100.0}}, "build": rl.source_stamp(ROOT)}
    cnt = {"aabb_tests": 1000, "prim_tests": 100, "shaded_hits": 10, "rays": 12, "draws": 99}
    a = rl.algorithmic_valu(costs, cnt, 5, 1e-6, root=ROOT)
    lanes = 20.0 * 1000 + 50.0 * 100 + 300.0 * 10 + 100.0 * 5
    assert a["lane_instructions_per_launch"] == lanes and abs(a["achieved"] - lanes / 1e-6 / 1e9) < 1e-9
    assert abs(a["peak"] - 256 * 4 * 64 * 2.4e9 / 2 / 1e9) < 1e-6 and abs(a["frac"] - a["achieved"] / a["peak"]) < 1e-12
    # twice the time, half the fraction; more overhead instructions change nothing (they are not in the formula)
    assert abs(rl.algorithmic_valu(costs, cnt, 5, 2e-6, root=ROOT)["frac"] - a["frac"] / 2) < 1e-12
    costs["build"]["digest"] = "0" * 40
    stale = rl.algorithmic_valu(costs, cnt, 5, 1e-6, root=ROOT)
    assert stale["frac"] is None and "stale_costs" in stale
    assert rl.algorithmic_valu(None, cnt, 5, 1e-6)["frac"] is None


def test_committed_valu_costs_have_the_four_operations(pkg):
    import json
    from rtiow_rust_amd import roofline as rl
    costs, rel = rl.load_valu_costs(ROOT)
    if costs is None:
        import pytest
        pytest.skip("no valu_costs entry in profiles/current.json yet")
    for kern in ("render_lean_pool", "render_full_pool", "render_full_pool2"):
        k = costs["kernels"][kern]
        assert 15 <= k["c_box"] <= 40 and 30 <= k["c_prim"] <= 150 and 200 <= k["c_shade"] <= 2000 and 100 <= k["c_cam"] <= 2000, (kern, k)
        for r in k["regions"].values():
            assert r["lines"][0] <= r["lines"][1] and r["valu"] <= r["instructions"]
