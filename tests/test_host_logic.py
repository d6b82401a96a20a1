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
from posegen_amd.dist import plan_tasks, render_path_distributed
from posegen_amd.render import render_path

class StubRenderer:                       # stands in for HipRenderer: a ray's maps = f(box, pose, camera, ray index)
    device = torch.device("cpu")
    rays = 0
    def set_chunk(self, c): self.chunk = c
    def render_frame_range(self, H, W, focal, c2w, box, skts, cyl, r0, r1, center=None, cam=None, **kw):
        assert r0 == 0 or r0 % self.chunk == 0, "a run starts on a nanmean group boundary"
        StubRenderer.rays += r1 - r0
        (tlx, tly), (brx, bry) = box
        v = (float(torch.as_tensor(skts).sum()) + float(np.asarray(c2w).sum())) % 1.0
        i = torch.arange(r0, r1, dtype=torch.float32)
        rgb = torch.stack([v + 0 * i, (i % 7) / 7, (i % 5) / 5], -1)
        disp = torch.full((r1 - r0,), float("nan"))        # empty rays: NaN disparity -> 0 (run_nerf.py:142-143)
        acc = 0.5 + 0.25 * torch.sin(i)
        return torch.cat([rgb.reshape(-1), disp, acc])
    def compose_frame(self, H, W, box, rgb_map, disp_map, acc_map, bg=None, base_bg=0., **kw):
        (tlx, tly), (brx, bry) = box
        rgb = torch.full((H, W, 3), float(base_bg)); disp = torch.zeros(H, W, 1); acc = torch.zeros(H, W, 1)
        bh, bw = bry - tly, brx - tlx
        rgb[tly:bry, tlx:brx] = rgb_map.view(bh, bw, 3) + (1 - acc_map.view(bh, bw, 1)) * base_bg
        disp[tly:bry, tlx:brx] = torch.nan_to_num(disp_map.view(bh, bw, 1), nan=0.0)
        acc[tly:bry, tlx:brx] = acc_map.view(bh, bw, 1)
        return rgb, disp, acc
    def render_frame(self, H, W, focal, c2w, box, skts, cyl, **kw):
        (tlx, tly), (brx, bry) = box
        n = (bry - tly) * (brx - tlx)
        p = self.render_frame_range(H, W, focal, c2w, box, skts, cyl, 0, n)
        StubRenderer.rays -= n
        return self.compose_frame(H, W, box, p[:3 * n].view(n, 3), p[3 * n:4 * n], p[4 * n:], base_bg=kw.get("base_bg", 0.))
class StubCaster:
    renderer = StubRenderer()
    module = property(lambda self: self)

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
H = W = 64
for F, hw in ((1, (H, W)), (3, (H, W)), (1, (np.int64(H), np.int64(W)))):     # F odd: a frame is cut between the two ranks
    _, kps, skts = syn.make_pose(F, 3)
    c2ws, focals = syn.make_camera(F, H, W)
    kw = dict(kp=torch.tensor(kps), skts=torch.tensor(skts), white_bkgd=True, ext_scale=0.001, ret_acc=True)
    rk = {"ray_caster": StubCaster(), "N_samples": 64, "N_importance": 16}
    want = render_path(torch.tensor(c2ws), (H, W, focals), 256, rk, **kw)
    before = StubRenderer.rays
    got = render_path_distributed(torch.tensor(c2ws), hw + (focals,), 256, rk, **kw)
    mine = StubRenderer.rays - before
    for a, b in zip(want[:3], got[:3]):
        assert a.shape == b.shape and np.array_equal(a, b), (rank, F)
    assert np.array_equal(np.array(got[4]), np.array(want[4]))
    total = sum(len(v) for v in want[3])
    n = torch.tensor([mine]); dist.all_reduce(n)
    assert int(n) == total, "every ray rendered exactly once across the ranks"
    assert abs(mine - total / 2) <= 256 + total // 50, ("balanced to about one group", mine, total)
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


def test_render_path_distributed_gloo_world2_cut_frames(tmp_path):
    """render_path_distributed == render_path on every rank: one frame on two ranks and three frames on two
    ranks (a frame's nanmean groups are cut between the ranks, every ray rendered once, loads balanced to a
    group), numpy-integer H / W included."""
    _run_world2(tmp_path, _GLOO_RENDER_WORKER)


def test_render_path_distributed_without_a_process_group_is_the_single_device_render():
    """No torch.distributed initialised: world 1, no collective, same frames as render_path."""
    ns = {}
    src = _GLOO_RENDER_WORKER.split("dist.init_process_group")[0].replace("sys.path.insert(0, sys.argv[1])", "")
    exec(src, ns)
    from posegen_amd.dist import render_path_distributed
    from posegen_amd.render import render_path
    from posegen_amd import synthetic as syn
    H = W = 48
    _, kps, skts = syn.make_pose(2, 3)
    c2ws, focals = syn.make_camera(2, H, W)
    kw = dict(kp=torch.tensor(kps), skts=torch.tensor(skts), white_bkgd=True, ext_scale=0.001, ret_acc=True)
    rk = {"ray_caster": ns["StubCaster"](), "N_samples": 64, "N_importance": 16}
    want = render_path(torch.tensor(c2ws), (H, W, focals), 256, rk, **kw)
    got = render_path_distributed(torch.tensor(c2ws), (H, W, focals), 256, rk, **kw)
    for a, b in zip(want[:3], got[:3]):
        assert np.array_equal(a, b)


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
    """pg_plan_frames (the work plan of the in-process multi-GPU renderer) and dist.plan_tasks (the same plan for
    one process per GPU): every ray of every frame in exactly one task, every cut on a multiple of `chunk` (the
    groups are those of the single-device render), one owner per frame, loads within about a group of the mean --
    also when the frame count is not a multiple of the worker count (20 frames on 8 GPUs)."""
    from posegen_amd.dist import plan_tasks
    rng = np.random.RandomState(0)
    for workers in (1, 2, 3, 8):
        for F in (1, 2, 5, 8, 9, 20):
            for chunk in (4096, 1000):
                n = [int(x) for x in rng.randint(1, 200000, size=F)]
                tasks = _plan(n, workers, chunk)
                assert tasks == [tuple(t) for t in plan_tasks(n, workers, chunk)], "the library and dist.py plan alike"
                for f in range(F):
                    runs = sorted((t[1], t[2]) for t in tasks if t[0] == f)
                    assert runs[0][0] == 0 and runs[-1][1] == n[f]
                    for (a0, a1), (b0, b1) in zip(runs, runs[1:]):
                        assert a1 == b0 and a1 % chunk == 0
                    owners = {t[4] for t in tasks if t[0] == f}
                    assert len(owners) == 1 and next(iter(owners)) in {t[3] for t in tasks if t[0] == f}
                assert all(0 <= t[3] < workers for t in tasks)
                load = [sum(t[2] - t[1] for t in tasks if t[3] == w) for w in range(workers)]
                mean = sum(n) / workers
                if mean >= 16 * chunk:                     # enough groups per worker for a balanced plan to exist
                    assert max(load) <= 1.05 * mean + chunk, (workers, F, chunk, load)
    # the GAN loop's call pattern: 20 frames on 8 devices (whole frames alone: 3:2 loads = 1.2 x the mean)
    n = [int(x) for x in np.random.RandomState(1).randint(70000, 130000, size=20)]
    for sizes in (n, [262144] * 20):
        tasks = _plan(sizes, 8, 4096)
        load = [sum(t[2] - t[1] for t in tasks if t[3] == w) for w in range(8)]
        assert max(load) <= 1.05 * sum(load) / 8
        assert sum(1 for t in tasks if t[1] == 0 and t[2] == sizes[t[0]]) >= 12, "most frames stay whole"
    # one frame, eight devices: eight runs of whole groups
    t = _plan([262144], 8, 4096)
    assert [x[2] - x[1] for x in t] == [32768] * 8 and [x[3] for x in t] == list(range(8))
    # frames that fit are never cut
    t = _plan([5000] * 8, 8, 4096)
    assert all(x[1] == 0 and x[2] == 5000 for x in t) and sorted(x[3] for x in t) == list(range(8))


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
    # beside the weak headline: the product's multi-GPU path (plan, all-gather, compose) on 20 culled frames
    ss = j["strong_scaling"]
    assert ss["scaling"] == "strong" and ss["frames_per_step"] == 20 and ss["n_gpus"] == 2 and ss["valid_rays_per_step"] > 0
    assert abs(ss["rays_per_s"] - ss["valid_rays_per_step"] / (ss["ms_per_step"] * 1e-3)) < 1e-6 * ss["rays_per_s"]
    # --scaling strong: that path IS the timed region; value = rays of the whole job / max-over-ranks time
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--res", "64", "--dry-run", "--scaling", "strong", "--frames", "5"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert j["scaling"] == "strong" and j["n_gpus"] == 2 and j["config"]["frames_per_step"] == 5
    assert "strong_scaling" not in j and j["value"] > 0
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


def test_checkpoint_discovery_follows_the_reference(tmp_path):
    """find_checkpoint = the reload rule of create_raycaster (core/raycasters.py:124-141): ft_path wins, else the
    last '*tar*' entry (sorted by name, 'pose' files skipped) of basedir/expname; none with no_reload."""
    from posegen_amd.raycaster import find_checkpoint
    d = tmp_path / "logs" / "exp"
    d.mkdir(parents=True)
    assert find_checkpoint(str(tmp_path / "logs"), "exp") is None
    for name in ("050000.tar", "150000.tar", "100000.tar", "200000_pose.tar", "args.txt"):
        (d / name).write_bytes(b"x")
    assert find_checkpoint(str(tmp_path / "logs"), "exp") == str(d / "150000.tar")
    assert find_checkpoint(str(tmp_path / "logs"), "exp", ft_path="None") == str(d / "150000.tar")
    assert find_checkpoint(str(tmp_path / "logs"), "exp", ft_path="/x/y.tar") == "/x/y.tar"
    assert find_checkpoint(str(tmp_path / "logs"), "exp", no_reload=True) is None


def test_unsupported_preproc_kwargs_are_refused_not_ignored():
    """a caller configured for something the kernels do not compute (another encoder, another
    density function or scale) gets an exception, not a ReLU / RelDist render.  Host logic only (no GPU)."""
    import pytest
    import torch.nn.functional as F
    from posegen_amd import surreal_config
    from posegen_amd.raycaster import HipRayCaster, _density_act

    class RelDistEncoder: pass
    class VecNormEncoder: pass
    class WorldToLocalEncoder: pass
    class RayAngEncoder: pass

    def caster(**kw):
        c = HipRayCaster.__new__(HipRayCaster)          # the check needs the configuration only
        c.cfg = surreal_config(**kw)
        return c
    ref_kwargs = {"pts_tr_fn": WorldToLocalEncoder(), "kp_input_fn": RelDistEncoder(), "view_input_fn": VecNormEncoder(),
                  "bone_input_fn": VecNormEncoder(), "density_scale": 1.0, "density_fn": F.relu}
    caster()._check_preproc_kwargs(ref_kwargs)          # what create_raycaster builds for the shipped configs: accepted
    caster()._check_preproc_kwargs({})
    with pytest.raises(NotImplementedError, match="RayAngEncoder"):
        caster()._check_preproc_kwargs(dict(ref_kwargs, view_input_fn=RayAngEncoder()))
    with pytest.raises(ValueError, match="density_scale"):
        caster()._check_preproc_kwargs(dict(ref_kwargs, density_scale=0.5))
    softplus = lambda x: F.softplus(x - 1.0, beta=1)    # get_density_fn's lambda (raycasters.py:233-236)
    with pytest.raises(NotImplementedError, match="density_fn"):
        caster()._check_preproc_kwargs(dict(ref_kwargs, density_fn=softplus))
    caster(density_type="softplus", softplus_shift=1.0)._check_preproc_kwargs(dict(ref_kwargs, density_fn=softplus))
    with pytest.raises(NotImplementedError, match="density_fn"):
        caster(density_type="softplus", softplus_shift=0.5)._check_preproc_kwargs(dict(ref_kwargs, density_fn=softplus))
    with pytest.raises(NotImplementedError, match="not supported"):
        caster()._check_preproc_kwargs({"subject_fn": object()})
    with pytest.raises(NotImplementedError, match="undefined"):
        _density_act("elu")


def test_box_pixel_ids_are_the_reference_valid_idxs_built_lazily():
    """rays.BoxPixelIds (the `valid_idxs` render_path returns) == kp_to_boxes' materialised ids, and the host-route
    frame_boxes == kp_to_boxes' boxes."""
    from posegen_amd import synthetic as syn
    from posegen_amd.rays import BoxPixelIds, frame_boxes, kp_to_boxes
    H = W = 96
    _, kps, _ = syn.make_pose(3, 11)
    c2ws, focals = syn.make_camera(3, H, W)
    cyls, bboxes, grids = kp_to_boxes(torch.tensor(c2ws), H, W, focals, kps=torch.tensor(kps), ext_scale=0.001)
    lazy = BoxPixelIds(bboxes, [g[3] for g in grids])
    assert len(lazy) == 3 and lazy._cache == {}
    assert lazy.counts() == [len(g[0]) for g in grids] and lazy._cache == {}
    for i, (rows, cols, h, w, *_rest) in enumerate(grids):
        assert torch.equal(lazy[i], rows * w + cols)
    assert sum(len(v) for v in lazy) == sum(lazy.counts())
    # list-likeness of the reference's return type: a plain list on request, concatenation, numpy, pickling
    import pickle
    as_list = lazy.tolist()
    assert isinstance(as_list, list) and all(torch.equal(a, b) for a, b in zip(as_list, lazy))
    assert isinstance(lazy + [torch.zeros(1)], list) and len(lazy + [torch.zeros(1)]) == 4 and len([1] + lazy) == 4
    arr = np.asarray(lazy)
    assert arr.dtype == object and arr.shape == (3,) and np.array_equal(arr[1], lazy[1].numpy())
    back = pickle.loads(pickle.dumps(lazy))
    assert isinstance(back, list) and all(torch.equal(a, b) for a, b in zip(back, lazy))
    c2, b2, meta = frame_boxes(object(), torch.tensor(c2ws), H, W, focals, kps=torch.tensor(kps), ext_scale=0.001)   # no pose_boxes: host route
    assert torch.equal(torch.as_tensor(c2), torch.as_tensor(cyls))
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(b2, bboxes))
    assert [m[:2] for m in meta] == [(H, W)] * 3


def test_hazard_audit_flags_a_transcendental_result_read_by_the_next_instruction(tmp_path):
    """tools/audit_asm_hazards.py (run by `make audit` on every fused kernel): on the gfx940 family a VALU instruction
    may not read the result of v_sin / v_cos / v_exp / v_rcp / v_rsq / v_sqrt / v_log in the very next issue slot; hipcc
    pads its own code but not inline asm (the first build of the compensated kernel's on-chip form read stale values in
    the first 16 lanes that way).  The audit must fail on the adjacent pair and pass once an instruction sits in between."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "audit_asm_hazards.py")
    head = "_Z6kernelv:\n\tv_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[4:7], a[0:15]\n"
    bad = head + "\tv_cos_f32_e32 v141, v194\n\tv_cvt_pk_f16_f32 v198, v140, v141\n\ts_endpgm\n"
    good = head + "\tv_cos_f32_e32 v141, v194\n\ts_nop 0\n\tv_cvt_pk_f16_f32 v198, v140, v141\n\ts_endpgm\n"
    chain = head + "\tv_rsq_f32_e32 v1, v2\n\tv_sin_f32_e32 v3, v1\n\tv_mul_f32_e32 v4, v5, v6\n\ts_endpgm\n"      # trans -> trans is not the hazard
    wide = head + "\tv_exp_f32_e32 v9, v2\n\tv_mfma_f32_32x32x16_f16 a[0:15], v[8:11], v[4:7], a[0:15]\n\ts_endpgm\n"  # a register range
    # the destination is a source too: accumulate opcodes, SDWA / DPP forms
    fmac = head + "\tv_exp_f32_e32 v5, v2\n\tv_fmac_f32_e32 v5, v6, v7\n\ts_endpgm\n"
    pkfmac = head + "\tv_rcp_f32_e32 v5, v2\n\tv_pk_fmac_f16 v5, v6, v7\n\ts_endpgm\n"
    dot = head + "\tv_sqrt_f32_e32 v5, v2\n\tv_dot2c_f32_f16 v5, v6, v7\n\ts_endpgm\n"
    sdwa = head + "\tv_log_f32_e32 v5, v2\n\tv_add_f32_sdwa v5, v6, v7 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\ts_endpgm\n"
    fmac_ok = head + "\tv_exp_f32_e32 v5, v2\n\tv_fmac_f32_e32 v8, v6, v7\n\ts_endpgm\n"
    for name, text, rc in (("bad", bad, 1), ("good", good, 0), ("chain", chain, 0), ("wide", wide, 1), ("fmac", fmac, 1),
                           ("pkfmac", pkfmac, 1), ("dot", dot, 1), ("sdwa", sdwa, 1), ("fmac_ok", fmac_ok, 0)):
        f = tmp_path / f"{name}.s"
        f.write_text(text)
        r = subprocess.run([sys.executable, tool, str(f)], capture_output=True, text=True)
        assert r.returncode == rc, (name, r.stdout)
