"""CPU tests of the host side: reference-mirroring geometry against the golden vectors,
the C-ABI surface of the shared library, the multi-GPU frame partition and its gather
(gloo, world_size 2)."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from posegen_amd import _ffi
from posegen_amd.dist import partition_frames
from posegen_amd.rays import get_rays, kp_to_valid_rays
from posegen_amd.skeleton import bones_to_pose, get_smpl_l2ws
from tests.helpers import load_golden

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kinematics_matches_reference_golden():
    g = load_golden("kinematics")
    kps, skts, l2ws = bones_to_pose(g["bones"], g["rest_pose"])
    np.testing.assert_allclose(l2ws, g["l2ws"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(kps, g["kps"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(skts, g["skts"], rtol=0, atol=1e-10)
    assert get_smpl_l2ws(g["bones"][0], g["rest_pose"]).dtype == np.float64


def test_valid_rays_matches_reference_golden():
    g = load_golden("valid_rays")
    H, W = int(g["H"]), int(g["W"])
    rays, vids, cyls, boxes = kp_to_valid_rays(torch.tensor(g["c2ws"]), H, W, g["focals"],
                                               kps=torch.tensor(g["kps"]), ext_scale=0.001)
    np.testing.assert_array_equal(cyls.numpy(), g["cyls"])
    for i in range(len(rays)):
        assert np.array_equal(np.array([boxes[i][0], boxes[i][1]]), g["boxes"][i])      # integer bbox: exact
        assert len(vids[i]) == int(g[f"n_valid_{i}"])
        assert np.array_equal(vids[i][:8].numpy(), g[f"vid_head_{i}"])
        assert np.array_equal(vids[i][-8:].numpy(), g[f"vid_tail_{i}"])
        np.testing.assert_array_equal(rays[i][0][:8].numpy(), g[f"rays_o_head_{i}"])
        np.testing.assert_array_equal(rays[i][1][:8].numpy(), g[f"rays_d_head_{i}"])
        np.testing.assert_array_equal(rays[i][1][-8:].numpy(), g[f"rays_d_tail_{i}"])


def test_get_rays_full_frame_equals_box_rays():
    g = load_golden("valid_rays")
    H, W = int(g["H"]), int(g["W"])
    rays, vids, _, _ = kp_to_valid_rays(torch.tensor(g["c2ws"][:1]), H, W, g["focals"][:1],
                                        kps=torch.tensor(g["kps"][:1]), ext_scale=0.001)
    ro, rd = get_rays(H, W, g["focals"][0], torch.tensor(g["c2ws"][0]))
    assert torch.equal(rd.reshape(-1, 3)[vids[0]], rays[0][1])


def test_library_exports_every_symbol_of_the_header():
    try:
        lib = _ffi.load_library()
    except _ffi.HipLibraryError as e:
        pytest.fail(f"library missing: {e} (run __graft_entry__.build())")
    hdr = open(os.path.join(REPO, "include", "posegen_hip.h")).read()
    declared = set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"pg_handle"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/posegen_hip.h but not exported"
        assert name in _ffi.PROTOTYPES, f"{name} has no ctypes prototype"
    assert lib.pg_abi_version() == _ffi.PG_ABI_VERSION


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_ffi.HipLibraryError):
        _ffi.load_library(str(tmp_path / "nope.so"))


def test_no_gpu_means_error_not_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from posegen_amd.config import surreal_config
    from posegen_amd.raycaster import HipRenderer
    with pytest.raises((_ffi.HipLibraryError, _ffi.PgError, RuntimeError)):
        HipRenderer(surreal_config(), "cuda:0")


def test_partition_frames_balanced_and_complete():
    rng = np.random.RandomState(0)
    for world in (1, 2, 3, 8):
        for F in (1, 2, 7, 20):
            n = rng.randint(100, 5000, size=F)
            parts = partition_frames(n, world)
            assert sorted(f for p in parts for f in p) == list(range(F))
            loads = [int(sum(n[f] for f in p)) for p in parts]
            if F >= world:
                assert max(loads) - min(loads) <= int(n.max())
    assert partition_frames([10] * 20, 8) == partition_frames([10] * 20, 8)


_GLOO_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from posegen_amd.dist import gather_frames, partition_frames
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_rays = [300, 100, 250, 50, 400]
parts = partition_frames(n_rays, world)
H = W = 4
mine = parts[rank]
local = torch.stack([torch.full((H, W, 5), float(f + 1)) for f in mine]) if mine else torch.zeros(0, H, W, 5)
full = gather_frames(local, mine, parts, len(n_rays))
assert full.shape == (5, H, W, 5)
for f in range(5):
    assert torch.all(full[f] == f + 1), (rank, f)
dist.barrier()
dist.destroy_process_group()
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"ok_{rank}"), "w").write("ok")
"""


def test_gather_frames_gloo_world2(tmp_path):
    _run_world2(tmp_path, _GLOO_WORKER)


_GLOO_RENDER_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from posegen_amd import synthetic as syn
from posegen_amd.dist import render_path_distributed
from posegen_amd.render import render_path

class StubRenderer:                       # stands in for HipRenderer: a frame = f(box, pose, camera)
    device = torch.device("cpu")
    calls = 0
    def set_chunk(self, c): self.chunk = c
    def render_frame(self, H, W, focal, c2w, box, skts, cyl, center=None, cam=None, **kw):
        StubRenderer.calls += 1
        (tlx, tly), (brx, bry) = box
        v = float(torch.as_tensor(skts).sum()) + float(np.asarray(c2w).sum())
        rgb = torch.full((H, W, 3), 1.0); disp = torch.zeros(H, W, 1); acc = torch.zeros(H, W, 1)
        rgb[tly:bry, tlx:brx] = v % 1.0
        disp[tly:bry, tlx:brx] = float("nan")      # empty rays: NaN disparity -> 0 (run_nerf.py:142-143)
        acc[tly:bry, tlx:brx] = (brx - tlx) / W
        return rgb, disp, acc
class StubCaster:
    renderer = StubRenderer()
    module = property(lambda self: self)

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
H = W = 64
for F in (1, 3):                          # F = 1 < world: rank 1 owns nothing and must still gather
    _, kps, skts = syn.make_pose(F, 3)
    c2ws, focals = syn.make_camera(F, H, W)
    kw = dict(kp=torch.tensor(kps), skts=torch.tensor(skts), white_bkgd=True, ext_scale=0.001, ret_acc=True)
    rk = {"ray_caster": StubCaster(), "N_samples": 64, "N_importance": 16}
    want = render_path(torch.tensor(c2ws), (H, W, focals), 4096, rk, **kw)
    before = StubRenderer.calls
    got = render_path_distributed(torch.tensor(c2ws), (H, W, focals), 4096, rk, **kw)
    mine = StubRenderer.calls - before
    for a, b in zip(want[:3], got[:3]):
        assert a.shape == b.shape and np.array_equal(a, b), (rank, F)
    assert np.array_equal(np.array(got[4]), np.array(want[4]))
    n = torch.tensor([mine]); dist.all_reduce(n)
    assert int(n) == F, "every frame rendered exactly once across the ranks"
dist.barrier()
dist.destroy_process_group()
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"ok_{rank}"), "w").write("ok")
"""


def _run_world2(tmp_path, source):
    script = tmp_path / "worker.py"
    script.write_text(source)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for attempt in range(3):        # the probed port can be taken again before torchrun binds it
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), REPO]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        if out.returncode == 0 or "address already in use" not in (out.stdout + out.stderr).lower():
            break
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()      # (stdout of the ranks interleaves)


def test_render_path_distributed_gloo_world2_fewer_frames_than_ranks(tmp_path):
    """render_path_distributed == render_path on every rank, including F = 1 on two ranks (the
    rank with an empty share must reach the all-gather instead of dying in torch.stack([]))."""
    _run_world2(tmp_path, _GLOO_RENDER_WORKER)


def _plan(n_rays, workers, chunk):
    import ctypes as C
    lib = _ffi.load_library()
    arr = (C.c_int64 * len(n_rays))(*n_rays)
    n = C.c_int()
    assert lib.pg_plan_frames(len(n_rays), arr, workers, chunk, None, 0, C.byref(n)) == 0
    out = (C.c_int32 * (5 * max(n.value, 1)))()
    assert lib.pg_plan_frames(len(n_rays), arr, workers, chunk, out, n.value, C.byref(n)) == 0
    return [tuple(out[5 * i:5 * i + 5]) for i in range(n.value)]      # (frame, r0, r1, worker, owner)


def test_frame_plan_covers_every_ray_once_on_group_boundaries():
    """pg_plan_frames (the work plan of the in-process multi-GPU renderer): whole frames by LPT when there
    are enough of them; otherwise a frame's nanmean groups are cut into contiguous runs, so every cut
    falls on a multiple of `chunk` and the groups are those of the single-device render."""
    rng = np.random.RandomState(0)
    for workers in (1, 2, 3, 8):
        for F in (1, 2, 5, 20):
            for chunk in (4096, 1000):
                n = [int(x) for x in rng.randint(1, 200000, size=F)]
                tasks = _plan(n, workers, chunk)
                for f in range(F):
                    runs = sorted((t[1], t[2]) for t in tasks if t[0] == f)
                    assert runs[0][0] == 0 and runs[-1][1] == n[f]
                    for (a0, a1), (b0, b1) in zip(runs, runs[1:]):
                        assert a1 == b0 and a1 % chunk == 0
                    owners = {t[4] for t in tasks if t[0] == f}
                    assert len(owners) == 1 and next(iter(owners)) in {t[3] for t in tasks if t[0] == f}
                assert all(0 <= t[3] < workers for t in tasks)
                if F >= workers:
                    assert all(t[1] == 0 and t[2] == n[t[0]] for t in tasks)          # whole frames
                    load = [sum(n[t[0]] for t in tasks if t[3] == w) for w in range(workers)]
                    assert max(load) - min(load) <= max(n)
                else:
                    used = {t[3] for t in tasks}
                    big = [f for f in range(F) if n[f] > chunk * workers]
                    if len(big) == F:
                        assert len(used) == workers                                    # nobody idles
    # one frame, eight devices: eight runs of whole groups
    t = _plan([262144], 8, 4096)
    assert [x[2] - x[1] for x in t] == [32768] * 8 and [x[3] for x in t] == list(range(8))


def test_bench_launches_itself_for_n_gpus_dry_run():
    """`python bench.py --gpus 2` with no rendezvous in the environment starts its own two
    ranks (as a child process) and relays rank 0's single JSON line; --dry-run swaps the renderer
    for a stub and RCCL for gloo, everything else (barrier, max over ranks, all-gather) is real."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--res", "64", "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["metric"].startswith("rendered rays/sec") and j["unit"] == "rays/s"
    assert abs(j["value"] - 2 * 64 * 64 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]
    # a child failure is an error exit, not a silent empty line
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--prec", "nope", "--dry-run"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0 and not bad.stdout.strip()


def test_inline_asm_ring_reads_are_not_touched_by_the_compiler():
    """The 16-bit kernel issues its ring reads by inline asm and retires them with counted
    waits; the compiled code must never copy / spill such a register in between."""
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    out = subprocess.run(["make", "-C", os.path.join(REPO, "posegen_amd", "csrc"), "audit"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "AUDIT OK" in out.stdout


def test_training_draws_follow_the_reference_places_and_test_mode():
    """make_training_draws: which numbers exist for which switches, their shapes and scales, and
    the pytest=True branch (numpy after seed 0, float64 -> float32, no B on the density noise:
    ray_utils.py:171-180, 241-244; nerf.py:179-182) equal to the numbers stored in the fixture
    made by the reference's own run."""
    from posegen_amd.raycaster import make_training_draws
    from tests.helpers import load_golden
    g = load_golden("rays_train")
    n, S, N = int(g["n_rays"]), int(g["n_samples"]), int(g["n_importance"])
    d = make_training_draws(n, S, N, perturb=1., raw_noise_std=float(g["raw_noise_std"]), ray_noise_std=0., pytest=True,
                            density_scale=7.0)
    assert set(d) == {"t_rand", "u_rand", "noise0", "noise1"}
    for k in d:
        assert np.array_equal(d[k].numpy(), g[k]), k
    assert make_training_draws(n, S, N) == {}
    assert set(make_training_draws(n, S, 0, perturb=1., raw_noise_std=1.)) == {"t_rand", "noise0"}
    torch.manual_seed(3)
    r = make_training_draws(4096, 8, 4, perturb=1., raw_noise_std=2., ray_noise_std=0.5, density_scale=3.0)
    assert r["t_rand"].shape == (4096, 8) and r["u_rand"].shape == (4096, 4) and r["ray_noise"].shape == (4096, 12, 3)
    assert 0 <= float(r["t_rand"].min()) and float(r["t_rand"].max()) < 1
    assert abs(float(r["noise0"].std()) - 6.0) < 0.15 and abs(float(r["noise1"].std()) - 6.0) < 0.15
    assert abs(float(r["ray_noise"].std()) - 0.5) < 0.01


def test_host_packer_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """The weight packer (pg_pack.cpp) is the host-side native code with real index arithmetic: every
    precision x program x frame-code combination under ASan + UBSan on the CPU (GPU sanitizers are not
    available on the pool)."""
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("ROCm clang++ not found")
    exe = str(tmp_path / "pack_asan")
    csrc = os.path.join(REPO, "posegen_amd", "csrc")
    build = subprocess.run([clang, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-fno-omit-frame-pointer", "-I", csrc, os.path.join(REPO, "tools", "sanitize", "pack_asan.cpp"),
                            os.path.join(csrc, "pg_pack.cpp"), "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitizer" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-3000:])
    assert "packer clean under ASan/UBSan" in run.stdout
