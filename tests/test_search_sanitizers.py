"""The threaded search driver under ThreadSanitizer and AddressSanitizer + UBSan (host code only; the GPU pool has
no sanitizer): the fork-join pool, the evaluation lane and the block arena run two depth-4 searches on four threads
and the sanitizers must have nothing to report.  The searches must have taken the wide path (batches of 320 nodes
or more: both worker groups of the pool, the tree update on the evaluation lane beside the hand-over of the move
scores and the arena copy), which the driver reports."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "batch_first_amd", "csrc", "search")


@pytest.mark.parametrize("flags", ["-fsanitize=thread", "-fsanitize=address,undefined -fno-sanitize-recover=undefined"])
def test_search_is_clean_under_host_sanitizers(tmp_path, flags):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "driver")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-I", SRC] + flags.split() + [
        os.path.join(ROOT, "tests", "search_sanitize_driver.cpp"), os.path.join(SRC, "search.cpp"),
        os.path.join(SRC, "board.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1")
    env.pop("LD_PRELOAD", None)   # a sanitizer runtime has to come first in the library list
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "WARNING: ThreadSanitizer" not in run.stderr and "ERROR: AddressSanitizer" not in run.stderr
    assert "runtime error" not in run.stderr
    lines = [l for l in run.stdout.splitlines() if l.startswith("value")]
    assert len(lines) == 2 and all(l.startswith("value 0.0 move ") for l in lines), lines   # material is level at depth 4
    for l in lines:
        words = l.split()
        wide, overlapped = int(words[words.index("wide") + 1]), int(words[words.index("overlapped") + 1])
        assert wide > 10 and overlapped > 10, l    # the wide path ran under the sanitizer


def test_pool_visits_every_index_once_under_thread_sanitizer(tmp_path):
    """The fork-join pool ends a job when its ITEMS are done and lets late workers find nothing (threads.h): many short
    jobs of changing size, narrow and wide, on eight threads under ThreadSanitizer; every index exactly once."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "pool_stress")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-I", SRC,
                    os.path.join(ROOT, "tests", "pool_stress_driver.cpp"), "-o", exe], check=True, capture_output=True, timeout=300)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe, "8", "20000"], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "WARNING: ThreadSanitizer" not in run.stderr
    assert run.stdout.strip().endswith("bad 0")
