"""profiles/traffic.json -- the rocprofv3 PMC figures bench.py quotes as `roofline.traffic` / `kernels_roofline.*.valu` -- must belong
to the kernels as they are: the commit it records (tools/summarise_profiles.py) has to contain the last change of the label-pass and
cube-pass sources.  Needs the git history (the driver's CPU run has it; a GPU box's snapshot has not: skipped there)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["kmeans-gpu_amd/csrc/kmg_table.hip", "kmeans-gpu_amd/csrc/kmg_cube.hip", "kmeans-gpu_amd/csrc/kmg_table.h",
                  "kmeans-gpu_amd/csrc/kmg_table_dev.h"]


def _git(*args):
    return subprocess.run(["git", "-C", ROOT, *args], capture_output=True, text=True)


def test_traffic_json_is_not_older_than_the_kernels_it_describes():
    if _git("rev-parse", "--git-dir").returncode != 0:
        pytest.skip("no git history here")
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    commit = t.get("commit")
    assert commit, "profiles/traffic.json records no commit: regenerate it with tools/profile_round.sh + tools/summarise_profiles.py"
    assert _git("cat-file", "-e", commit + "^{commit}").returncode == 0, f"traffic.json names an unknown commit {commit}"
    last = _git("log", "-1", "--format=%H", "--", *KERNEL_SOURCES).stdout.strip()
    assert last, "no commit touches the kernel sources?"
    ok = _git("merge-base", "--is-ancestor", last, commit).returncode == 0
    assert ok, (f"{KERNEL_SOURCES} changed in {last[:10]} after the profile of {commit[:10]} (tag {t.get('profile_tag')}): the counter "
                "figures of the bench line are stale -- run tools/profile_round.sh on the GPU box and tools/summarise_profiles.py")
