"""Headline benchmark: rays/s (and samples/s) on the Book-2 final scene, 800x800, max depth 50, on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             (ranks started by torchrun)

One step = one render pass: --spp samples per pixel per GPU in the job (N * spp per pixel over the whole 800x800
image, tiles dealt round-robin to the ranks), i.e. a fixed number of paths per GPU per step (weak scaling). Inputs
(scene blob, camera, framebuffer) are resident in HBM before the timed region. Rank 0 prints one JSON line.

ray = one path segment = one world.hit issued by ray_color (main.rs:56), primary rays included;
sample = one iteration of the reference's sample loop (main.rs:161-167).
"""
import argparse
import hashlib
import json
import os
import struct
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X vector FP64 spec peak
# Algorithmic HBM bytes k_extend_tree moves per ray segment in this layout: ray record read (7 x f64 = 56) + hit record
# write (16). (Path i of a bounce is record i: no queue index is read.)
EXTEND_BYTES_PER_SEGMENT = 72
# Whole-path compulsory path-state traffic per segment, f64 layout: extend 72; shade reads 56 (ray) + 16 (hit) + 24
# (throughput) and writes 56 (next ray) + 24; + 8 for the slot / pixel words that ride in the 64-byte ray records,
# + 24 / rays-per-sample for the radiance written once per path and read once by the accumulation (~ 10).
PATH_BYTES_PER_SEGMENT = 266
# Work per segment under the REFERENCE's traversal (always both children, no ordering), counted by the oracle and
# committed as tests/golden/visits.json: flops = 27 per box test (3 divides, 6 sub, 6 mul, 6 select, 6 compare) + 25 per
# sphere + 12 per rect + 60 for the scatter; scene bytes = 32 per box test and sphere, 48 per rect.
FLOPS = dict(node_tests=27, sphere_tests=25, moving_sphere_tests=30, rect_tests=12, medium_tests=60)
SCENE_BYTES = dict(node_tests=32, sphere_tests=32, moving_sphere_tests=80, rect_tests=48, medium_tests=40)
SHADE_FLOPS = 60


def reference_work_per_segment(scene):
    path = os.path.join(ROOT, "tests", "golden", "visits.json")
    if not os.path.exists(path):
        return None
    v = json.load(open(path)).get(scene)
    if not v:
        return None
    seg = float(v["ray_segments"])
    per = {k: v[k] / seg for k in FLOPS}
    return {"per_segment": per, "flops": SHADE_FLOPS + sum(FLOPS[k] * per[k] for k in FLOPS),
            "scene_bytes": sum(SCENE_BYTES[k] * per[k] for k in FLOPS)}


def native_oracle():
    """The oracle compiled -O3 -march=native ON THIS HOST (the shipped library is built for any x86-64, on another
    machine) -> path of the library, or None when there is no compiler. About 15 s, once per bench run."""
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "oracle", "oracle.cpp")
    out = os.path.join(tempfile.mkdtemp(prefix="rt_oracle_native_"), "liboracle_native.so")
    cmd = ["g++", "-O3", "-march=native", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", src, "-o", out]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    except (OSError, subprocess.SubprocessError):
        return None
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(desc, info, seed, target_seconds, scene_name="final_scene"):
    """The oracle (a restatement of the reference's CPU renderer, not the Rust binary) timed on ALL of this host's
    cores over a bounded sample of the same workload: whole 800x800 image, as many spp as fit in about target_seconds."""
    native = native_oracle()
    if native:
        os.environ["RT_ORACLE_SO"] = native
    import checkers as ck
    from raytracer_amd import api
    cores = max(1, len(os.sched_getaffinity(0)))
    osc = ck.OracleScene(desc)
    p = api.params_from_info(info, seed=seed)
    _, st = osc.render(info.camera, p, 1, threads=cores)                       # calibration: 1 spp
    spp = int(max(1, min(256, target_seconds / max(st.seconds, 1e-3))))
    _, st = osc.render(info.camera, p, spp, sample_offset=1, threads=cores)
    # and as the reference runs by default: 2 worker threads over 16 row bands (main.rs:128,143-152), 1 spp
    _, st2 = osc.render(info.camera, p, 1, sample_offset=1 + spp, threads=2, jobs=16)
    return {
        "value": st.ray_segments / st.seconds, "unit": "rays/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
        "build": "g++ -O3 -march=native -ffp-contract=off, compiled on this host" if native else
                 "g++ -O3 -ffp-contract=off (generic x86-64 build shipped with the tree; no compiler on this host)",
        "samples_per_sec": st.camera_samples / st.seconds,
        "sample": "oracle (C++ f64 restatement of the reference renderer, not the Rust binary; %d threads, %d row-band jobs), "
                  "%s %dx%d, %d spp, depth %d: %.1f s" % (st.threads, st.jobs, scene_name, info.width, info.height, spp, info.max_depth, st.seconds),
        "reference_default": {"value": st2.ray_segments / st2.seconds, "unit": "rays/s", "cores": 2,
                              "sample": "the same with the reference's default 2 workers / 16 jobs, 1 spp: %.1f s" % st2.seconds},
    }


def launch_ranks(world):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks (fresh children; nothing here has
    touched a GPU), lets rank 0's JSON line through and exits with the worst exit code."""
    have = torch.cuda.device_count()  # counts devices without initialising one
    if "--rehearse-on-one-gpu" in sys.argv:
        have = world if have >= 1 else 0
    if have < world:
        raise SystemExit("bench.py --gpus %d needs %d GPUs on this machine; it has %d" % (world, world, have))
    from raytracer_amd import dist as rdist
    rc, codes = rdist.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], world, timeout=float(os.environ.get("RT_BENCH_TIMEOUT", "3300")))
    if rc != 0:
        print("bench.py: rank exit codes %s" % codes, file=sys.stderr)
    sys.exit(rc if 0 <= rc < 256 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp", type=int, default=1536, help="samples per pixel per GPU per step (reduced - 1536, 1024, 512, ... - until the path state fits the GPU)")
    ap.add_argument("--scene", default="final_scene", help="final_scene (the metric's configuration), cornell_box, random_scene, stress_volume, "
                                                           "spheres_grid, my_scene, cornell_smoke, ...")
    ap.add_argument("--n", type=int, default=0, help="object count of scaled scenes (spheres_grid: default 1 000 000)")
    # the workload constants of the reference's main() (main.rs:75-79) and the render's own switches
    ap.add_argument("--width", type=int, default=0, help="image width (default: the scene's own configuration)")
    ap.add_argument("--height", type=int, default=0, help="image height")
    ap.add_argument("--depth", type=int, default=0, help="max_depth (main.rs:79; default: the scene's, 50)")
    ap.add_argument("--seed", type=lambda v: int(v, 0), default=0x5EED0002, help="render seed (Philox key material)")
    ap.add_argument("--compat", type=int, default=0, help="bit 0: book-correct RotateY back-rotation, bit 1: book-correct XY lens disk "
                                                          "(default 0: as the reference computes them)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="every rank renders on cuda:0 and the collectives run over gloo: the N-rank code path (launch, "
                         "broadcast, tile sharding, gather) on a machine with one GPU; the rates it prints mean nothing")
    args = ap.parse_args()

    world = args.gpus
    if world < 1:
        raise SystemExit("--gpus must be at least 1")
    if world > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(world)  # does not return

    from raytracer_amd import api
    from raytracer_amd import dist as rdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        if int(os.environ["WORLD_SIZE"]) != world:
            raise SystemExit("--gpus %d but the launcher started %s ranks" % (world, os.environ["WORLD_SIZE"]))
        if args.rehearse_on_one_gpu:
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        rccl_ranks = dist.get_world_size()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    scene_seed, render_seed = 0x5EED0001, args.seed
    ctx = api.Context(local_rank)
    desc = None
    if rank == 0:
        b = api.Builder(seed=scene_seed)
        b.register_reference_textures(tuple(api.TEXTURE_FILES) if args.scene == "my_scene" else ("earthmap.jpg",))
        info = b.named_scene(args.scene, scene_seed, args.n)
        blob = b.compile(info.world)
        desc = b.describe(info.world)
    if world > 1:
        # the scene is compiled once and broadcast over RCCL; camera/image parameters are tiny and rebuilt per rank
        buf = rdist.broadcast_blob(blob if rank == 0 else None, device)
        if rank != 0:
            b = api.Builder(seed=scene_seed)
            b.register_reference_textures(tuple(api.TEXTURE_FILES) if args.scene == "my_scene" else ("earthmap.jpg",))
            info = b.named_scene(args.scene, scene_seed, args.n)
        scene = ctx.upload_from_device(buf.data_ptr(), buf.numel())
    else:
        scene = ctx.upload(blob)

    if args.width or args.height:
        info.width, info.height = args.width or info.width, args.height or info.height
    if args.depth:
        info.max_depth = args.depth
    W, H = info.width, info.height
    # One step = one pass of spp_step samples per pixel: the 50 bounces of a pass are 50 dependent rounds of launches, each
    # with a latency floor, so a pass is made as large as the HBM takes (232 B of path state per path: 1536 spp of this
    # GPU's share of 800x800 pixels = 228 GB). The first pass allocates; if it does not fit, every rank steps down together.
    while True:
        spp_step = args.spp * world
        params = api.params_from_info(info, seed=render_seed, tile=rdist.DEFAULT_TILE, tile_rank=rank, tile_world=world,
                                      spp_per_pass=spp_step, profile=1, compat=args.compat)
        d_sum = torch.zeros(W * H * 3, dtype=torch.float64, device=device)
        torch.cuda.synchronize()  # the library launches on its own stream: the buffer must be zeroed before it is handed over
        fits = torch.ones(1, dtype=torch.int32, device=device)
        try:
            ctx.render_accumulate(scene, info.camera, params, 0, spp_step, d_sum.data_ptr(), api.rt_stats())
        except api.RtError as e:
            if e.code != -4:  # RT_ERR_OOM
                raise
            fits.zero_()
        if world > 1:
            dist.all_reduce(fits, op=dist.ReduceOp.MIN)
        if int(fits.item()):
            break
        if args.spp <= 16:
            raise SystemExit("path state does not fit the GPU even at 16 spp per step")
        args.spp = args.spp * 2 // 3 if args.spp % 3 == 0 else args.spp // 2
    d_sum.zero_()
    torch.cuda.synchronize()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    step_index = 0
    warm = api.rt_stats()
    for _ in range(args.warmup):
        ctx.render_accumulate(scene, info.camera, params, step_index * spp_step, spp_step, d_sum.data_ptr(), warm)
        step_index += 1
    stats = api.rt_stats()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.render_accumulate(scene, info.camera, params, step_index * spp_step, spp_step, d_sum.data_ptr(), stats)
        step_index += 1
    sync()
    elapsed = time.perf_counter() - t0

    totals = torch.tensor([float(stats.ray_segments), float(stats.camera_samples), elapsed], dtype=torch.float64, device=device)
    if world > 1:
        t_max = totals[2:].clone()
        dist.all_reduce(totals[:2], op=dist.ReduceOp.SUM)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        elapsed = float(t_max.item())
        image = rdist.gather_image(d_sum, W, H, rdist.DEFAULT_TILE, rank, world)
    else:
        image = d_sum.view(H, W, 3)
    segments, samples = float(totals[0].item()), float(totals[1].item())

    if rank == 0:
        img = image.cpu().numpy()
        seg_per_launch = stats.extend_segments / max(1, stats.extend_launches)
        avg_ms = stats.extend_ms / max(1, stats.extend_launches)
        achieved = EXTEND_BYTES_PER_SEGMENT * seg_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        ref_work = reference_work_per_segment(args.scene)
        rate = segments / elapsed / world  # per GPU
        # Counters cannot be read inside this process: they come from rocprofv3 --pmc runs (tools/gpu_profile.sh), kept in
        # profiles/ with the scene, image and pass size they were taken at (counters_from). They are quoted only for the
        # workload they were measured on: another scene or image size gets null, not Book-2's figures.
        def counters_of(path):
            if not os.path.exists(path):
                return None
            j = json.load(open(path))
            same = (j.get("scene", "final_scene") == args.scene and j.get("width", 800) == W and j.get("height", 800) == H and
                    j.get("max_depth", 50) == info.max_depth and args.compat == 0)
            return j if same else None
        traffic, counters_from, valu = None, {}, None
        # the closest-hit kernel of this scene: a scene whose world is all on the lock-step list (RtBlobHeader::world_tree,
        # the u32 at byte 24, is 0) runs k_extend_list, every other one k_extend_tree
        extend_kernel = "k_extend_list" if struct.unpack_from("<I", bytes(blob[:32]), 24)[0] == 0 else "k_extend_tree"
        def first_of(*names):  # per-scene files (tools/gpu_profile_config.sh), then the bench scene's (tools/gpu_profile.sh)
            for nm in names:
                j = counters_of(os.path.join(ROOT, "profiles", nm))
                if j:
                    return nm, j
            return None, None
        pmc_name, pmc = first_of(args.scene + "_extend_hbm_traffic.json", "extend_hbm_traffic.json")
        if pmc:
            traffic = pmc.get("hbm_bytes_per_launch_corrected")
            counters_from["traffic"] = {"file": "profiles/" + pmc_name, "spp_per_pass": pmc.get("spp_per_pass"),
                                        "segments_per_launch": pmc.get("segments_per_launch"),
                                        "correction": "FETCH_SIZE x 2 (gfx950, 16-byte-per-lane streaming reads) + WRITE_SIZE"}
        sq_name, sqj = first_of(args.scene + "_sq_counters.json", "sq_counters.json")  # rocprofv3 --pmc SQ_* passes
        if sqj:
            sq = sqj.get(extend_kernel) or sqj.get("k_extend_tree", {})
            if sq.get("valu_busy") is not None and sq.get("lane_use") is not None:
                # the bound this kernel runs against: the share of the vector unit's issue slots that do useful lanes
                valu = {"busy": sq["valu_busy"], "lane_use": sq["lane_use"], "frac": sq["valu_busy"] * sq["lane_use"],
                        "unit": "share of VALU issue slots x lanes (1 = every SIMD issues a full-width instruction every 4 cycles)"}
            counters_from["valu"] = {"file": "profiles/" + sq_name, "spp_per_pass": sqj.get("spp_per_pass")}
        if not counters_from:
            counters_from = {"note": "no rocprofv3 counters are kept for this scene / image size / depth: traffic and valu are null"}
        out = {
            "metric": "rays_per_sec", "value": segments / elapsed, "unit": "rays/s",
            "samples_per_sec": samples / elapsed,
            "n_gpus": world, "rccl_ranks": rccl_ranks, "collectives": "gloo (one-GPU rehearsal)" if args.rehearse_on_one_gpu else "rccl", "rays_per_sec_per_gpu": rate, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, %dx%d, max depth %d, %d spp per step (%d per GPU), scene seed 0x5EED0001, render seed 0x5EED0002" %
                                   ("Book-2 final scene (final_scene, main.rs:576-688)" if args.scene == "final_scene" else "scene " + args.scene,
                                    W, H, info.max_depth, spp_step, args.spp) + (", render seed %#x" % render_seed if render_seed != 0x5EED0002 else "") +
                                   (", compat %d" % args.compat if args.compat else ""),
                       "scene": args.scene, "width": W, "height": H, "spp_per_step": spp_step, "max_depth": info.max_depth,
                       "tile": rdist.DEFAULT_TILE, "sharding": "tiles round-robin over %d rank(s)" % world},
            "rays_per_sample": segments / samples,
            "image_sha256": hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest(),
            "image_mean": float(img.mean() / (step_index * spp_step)),
            "roofline": {
                # achieved / peak / frac: the kernel's ALGORITHMIC path-state bytes over its HIP-event time against the HBM
                # peak (measured here). The kernel is not bound by HBM: the scene sits in L2 / LDS, and what limits it is
                # the vector unit under divergence - valu_busy x valu_lane_use, from the SQ counters (counters_from) - and the
                # chain of dependent steps a wave runs per node and leaf, with four waves per SIMD to overlap them.
                "kernel": extend_kernel, "bound": "valu", "valu": valu,
                # the HBM side, as the bench contract states it (achieved / peak / frac are HBM figures, not the bound's):
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "frac_of": "hbm", "traffic": traffic,
                "bytes_per_segment": EXTEND_BYTES_PER_SEGMENT, "segments_per_launch": seg_per_launch,
                "avg_launch_ms": avg_ms, "launches": int(stats.extend_launches),
                "counters_from": counters_from,
                "note": "bound by VALU issue and the latency of each wave's dependent steps under divergence, not by HBM: "
                        "valu.busy counts every vector instruction as 4 SIMD cycles, valu.frac = busy x lane_use is the figure "
                        "for the stated bound; achieved / peak / frac are the kernel's compulsory path-state bytes against "
                        "the HBM peak; traffic = HBM bytes per launch from FETCH_SIZE x 2 + WRITE_SIZE at the pass size named in "
                        "counters_from; valu and traffic are null when no counters are kept for this workload",
            },
            "path_hbm_frac": PATH_BYTES_PER_SEGMENT * rate / 1e9 / HBM_PEAK_GBS,
            # the same rate priced in the reference traversal's arithmetic and scene reads (the GPU path does less of both)
            "reference_work": None if ref_work is None else dict(
                ref_work, fp64_vector_frac=ref_work["flops"] * rate / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                scene_read_gbs=ref_work["scene_bytes"] * rate / 1e9),
            "kernel_ms": {"raygen": stats.raygen_ms, "extend_tree": stats.extend_ms, "shade_main": stats.shade_main_ms,
                          "accumulate": stats.accumulate_ms,
                          "shade_textured_lambertian": stats.shade_ms[0],
                          "shade_textured_diffuse_light": stats.shade_ms[3],
                          "shade_textured_isotropic": stats.shade_ms[4]},
        }
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(desc, info, render_seed, args.cpu_seconds, args.scene)
        print(json.dumps(out), flush=True)

    scene.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
