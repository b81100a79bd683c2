"""The host-side code under AddressSanitizer + UndefinedBehaviorSanitizer: the scene loader, the tree
builder and flattener, the device stage bodies compiled for the host (tests/hostsim) and the CPU
oracle, linked into one executable (tests/hostsim/sanitize_main.cpp; `make -C oracle asan` or
`make -C tests/hostsim asan`).  GPU code cannot run under a sanitizer here; the stage bodies are the
same source the kernels wrap."""
import os
import subprocess

import pytest

import edge_scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "tests", "hostsim", "_build", "sanitize_check")
ENV = dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module")
def sanitize_check():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    return BINARY


def run(binary, w, h, n, paths):
    out = subprocess.run([binary, str(w), str(h), str(n)] + list(paths), env=ENV, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
    return out.stdout


def test_scene_files_under_sanitizers(sanitize_check, scene_path):
    text = run(sanitize_check, 80, 60, 2, [scene_path("mixed_features"), scene_path("bunny_small"), scene_path("mixed_ortho"),
                                            scene_path("cornell_room")])
    assert text.count("primary ids equal") == 4 * 2 * 2 * 2  # scenes x sample patterns x modes x partitions


@pytest.mark.parametrize("name", edge_scenes.NAMES)
def test_corner_case_scenes_under_sanitizers(sanitize_check, tmp_path, name):
    """Odd sizes (partly filled tiles), one pixel, no lights, a camera inside a box, a mesh with a
    zero-area triangle, a mesh whose leaves are flat: the index arithmetic at its edges."""
    scene_text, data_dir, w, h, n = edge_scenes.scenes(str(tmp_path))[name]
    path = os.path.join(data_dir or str(tmp_path), f"{name}.json")
    with open(path, "w") as f:
        f.write(scene_text)
    run(sanitize_check, w, h, n, [path])
