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
