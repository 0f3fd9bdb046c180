"""The tile plan of the communication-avoiding sweep program (control_amd/csrc/tiles.cpp) and
the scheme of its kernel (rings, credit, hand-offs of two iterates), emulated on the CPU by
tests/native/tile_emu.cpp and compared bit for bit with the plain recurrence.  Host-only: the
GPU kernel itself is compared with the plain launches in tests/test_gpu_parity.py."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "tile_emu")


@pytest.fixture(scope="module")
def emu():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(
        ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__",
         "-I/opt/rocm/include", os.path.join(ROOT, "tests", "native", "tile_emu.cpp"),
         os.path.join(ROOT, "control_amd", "csrc", "tiles.cpp"), "-o", EXE,
         "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


# nx, ny, tiles, depth (0: modelled), threads, its, levels
CASES = [(33, 29, 12, 0, 64, 11, 3),       # ragged tile sizes, modelled depth
         (64, 50, 16, 3, 128, 9, 4),       # depth does not divide the step count
         (40, 40, 7, 1, 256, 5, 3),        # one step per hand-off, odd tile count
         (25, 25, 4, 6, 256, 2, 3),        # rings cover the whole mesh; two-step solves
         (257, 257, 256, 0, 512, 20, 2)]   # BASELINE configs[1] mesh, one tile per CU


def test_uncoupled_components_get_tiles_of_their_own(emu):
    """A vector-valued block whose components do not couple (P2 velocity blocks: two copies of
    the scalar graph, interleaved node by node) is partitioned component by component: tiles made
    of pieces of both copies had 256 ring rows around 127 own rows on this mesh, 67 now."""
    r = subprocess.run([emu, "129", "129", "256", "1", "512", "6", "2", "1", "2"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches: 0 of" in r.stdout
    halo = int(r.stdout.split("halo")[1].split()[0])
    assert halo <= 80, r.stdout


def test_wide_rows_of_a_3d_mesh(emu):
    """15-point structure of Kuhn cubes (3-D P1), Dirichlet faces outside the tiles, depth 2 with
    three row slots: the scheme on rows wider than the 2-D ones and rings that outgrow the tiles."""
    r = subprocess.run([emu, "17", "17", "24", "2", "128", "7", "3", "1", "1", "13"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches: 0 of" in r.stdout


def test_tiles_from_coordinates(emu):
    """With the rows' coordinates the parts are boxes (recursive coordinate bisection): same
    scheme, smaller rings than the graph bisection's slanted parts -- 257^2 at depth 7: 630
    against 784 ring rows; 65^3 at depth 1: 760 against 1 030 (33^3 in 64 tiles: 448 against 512); the components of a vector-valued block share
    their nodes' coordinates and get tiles of their own (54 ring rows; 76 in tiles of both)."""
    for args, bound in ((["257", "257", "256", "7", "1024", "4", "2", "1", "1", "1", "1"], 640),
                        (["33", "33", "64", "1", "512", "4", "2", "1", "1", "33", "1"], 460),
                        (["129", "129", "256", "1", "512", "6", "2", "1", "2", "1", "1"], 60)):
        r = subprocess.run([emu] + args, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "mismatches: 0 of" in r.stdout
        assert int(r.stdout.split("halo")[1].split()[0]) <= bound, r.stdout


@pytest.mark.parametrize("case", CASES)
def test_tile_scheme_is_bit_identical_to_plain_recurrence(emu, case):
    r = subprocess.run([emu] + [str(c) for c in case], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches: 0 of" in r.stdout
