#!/usr/bin/env python3
"""Two ranks on ONE GPU over gloo, CoinTrainer: per-step digests of student / CKG weights and of the reduced gradients, to find where
two ranks diverge.  Launch: python tools/two_rank_probe.py [KEY=VALUE cfg overrides ...]  (spawns both ranks itself).
Env knobs forwarded to the ranks: COIN_STEP_GRAPHS, COIN_REDUCER_RESLICE, PROBE_BURNED, PROBE_STEPS."""
import hashlib
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def rank_main():
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    single = os.environ.get("PROBE_SINGLE")   # "nccl" / "gloo": ONE rank, the collectives forced on (coin_amd.parallel._FORCE)
    if single == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    if single:
        import coin_amd.parallel as PAR

        PAR._FORCE[0] = True
        os.environ["COIN_GRAD_ARENA"] = "1"
    rank = dist.get_rank()
    if os.environ.get("PROBE_DUMP_AFTER"):   # a hang: every thread's Python stack to stderr after this many seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["PROBE_DUMP_AFTER"]), repeat=False, file=sys.stderr)
    from coin_amd import graphs as G
    from coin_amd.config import get_cfg
    from coin_amd.data.synthetic import synthetic_offline_detections
    from coin_amd.engine import CoinTrainer

    burned = os.environ.get("PROBE_BURNED", "0") == "1"
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "coin", "GDINO", "foggy_synthetic.yaml"))
    over = ["SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.NUM_IMAGES", 2, "AMD.SYNTHETIC.HEIGHT", 384, "AMD.SYNTHETIC.WIDTH", 640,
            "AMD.TEXT_TEMPLATES", 2, "MODEL.DEVICE", "cuda:0", "CLOUD.BURN_UP_STEP", 0 if burned else 100,
            "CLOUD.OFFLINE_TEACHER_UPDATE_ITER", 1, "CLOUD.PROTOTYPE_UPDATE_START", 0, "CLOUD.CLS_B_THRESH", 0.2]
    for kv in sys.argv[2:]:
        k, v = kv.split("=", 1)
        over += [k, json.loads(v)]
    cfg.merge_from_list(over)
    G.set_enabled(os.environ.get("COIN_STEP_GRAPHS", "1") != "0")
    torch.manual_seed(11)
    tr = CoinTrainer(cfg)
    real_forward, g_det = tr.offline_teacher.forward, torch.Generator().manual_seed(7 + rank)

    def teacher(batched_inputs, branch=None, **kw):
        real_forward(batched_inputs, branch=branch, **kw)
        return [synthetic_offline_detections(tr.model_CLOUD.entry(d["file_name"]), g_det, device="cuda:0") for d in batched_inputs]

    tr.offline_teacher.forward = teacher

    def digest(ts):
        h = hashlib.sha256()
        for t in ts:
            h.update(t.detach().float().cpu().numpy().tobytes())
        return h.hexdigest()[:12]

    names = {id(p): n for n, p in list(tr.model.named_parameters()) + [("merge." + n, p) for n, p in tr.merge.named_parameters()]}
    real_step = tr.optimizer.step
    grads = {}

    def step(*a, **k):
        torch.cuda.synchronize()
        grads["g"] = {names.get(id(p), "?"): digest([p.grad]) for p in tr.optimizer.params if p.grad is not None}
        grads["scale"] = repr(k.get("inv_loss_scale"))
        return real_step(*a, **k)

    tr.optimizer.step = step
    if os.environ.get("PROBE_LOG_COLLECTIVES"):   # the order of the collectives per rank (they are matched by order)
        import threading

        real_ar, real_bc = dist.all_reduce, dist.broadcast
        seq = [0]

        def ar(t, *a, **k):
            seq[0] += 1
            print(f"[rank {rank}] coll {seq[0]}: all_reduce {t.numel()} thread={threading.current_thread().name}", file=sys.stderr, flush=True)
            return real_ar(t, *a, **k)

        def bc(t, *a, **k):
            seq[0] += 1
            print(f"[rank {rank}] coll {seq[0]}: broadcast {t.numel()}", file=sys.stderr, flush=True)
            return real_bc(t, *a, **k)

        dist.all_reduce, dist.broadcast = ar, bc
    out = []
    import time

    if os.environ.get("PROBE_WATCHDOG"):   # every 10 s: is the device idle, which collectives are still open
        import threading

        def watch():
            while True:
                time.sleep(10)
                try:
                    idle = torch.cuda.default_stream(0).query()
                    open_s = [(i, s.work.is_completed()) for i, s in enumerate(tr.reducer.slices) if s.work is not None]
                    open_m = [(i, s.work.is_completed()) for i, s in enumerate(tr.reducer_merge.slices) if s.work is not None]
                    print(f"[rank {rank}] watchdog: default stream idle={idle} student works={open_s} merge works={open_m} next={tr.reducer._next}", file=sys.stderr, flush=True)
                except Exception as ex:
                    print(f"[rank {rank}] watchdog: {type(ex).__name__}: {ex}", file=sys.stderr, flush=True)

        threading.Thread(target=watch, daemon=True).start()

    for i in range(int(os.environ.get("PROBE_STEPS", "5"))):
        t0 = time.time()
        tr.run_step()
        t1 = time.time()
        tr.prepare_next()
        torch.cuda.synchronize()
        print(f"[rank {rank}] step {i}: run_step {t1 - t0:.2f} s, prepare_next + sync {time.time() - t1:.2f} s", file=sys.stderr, flush=True)
        out.append({"step": i, "student": digest(tr.optimizer.params), "merge": digest(tr.optimizer_merge.params), "grads": dict(grads["g"]), "scale": grads["scale"]})
    dist.barrier()
    dist.destroy_process_group()
    print("RESULT " + json.dumps({"rank": rank, "steps": out, "stats": dict(G.STATS)}))


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    world = 1 if os.environ.get("PROBE_SINGLE") else 2
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", LOCAL_WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank"] + sys.argv[1:], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=float(os.environ.get("PROBE_TIMEOUT", "900")))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            for r, q in enumerate(procs):
                o, e = q.communicate()
                print(f"TIMEOUT; stderr tail of rank {r}:\n" + "\n".join(l for l in e[-9000:].splitlines() if "UserWarning" not in l and "amdgpu.ids" not in l))
            sys.exit(2)
        print("\n".join(l for l in e.splitlines() if l.startswith("[rank")))
        if os.environ.get("PROBE_DUMP_AFTER"):
            print(e[-6000:])
        if p.returncode != 0:
            print(o[-1500:], e[-3000:])
            sys.exit(1)
        res.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]))
    if len(res) == 1:
        print("single rank finished", len(res[0]["steps"]), "steps;", res[0]["stats"])
        return
    a, b = sorted(res, key=lambda d: d["rank"])
    for sa, sb in zip(a["steps"], b["steps"]):
        diff = sorted(n for n in sa["grads"] if sa["grads"][n] != sb["grads"].get(n))
        print(f"step {sa['step']}: student {'==' if sa['student'] == sb['student'] else '!='}  merge {'==' if sa['merge'] == sb['merge'] else '!='}  "
              f"scale {sa['scale']} / {sb['scale']}  reduced gradients that differ: {len(diff)} of {len(sa['grads'])} {diff[:8]}")
    print("stats", a["stats"])


if __name__ == "__main__":
    rank_main() if len(sys.argv) > 1 and sys.argv[1] == "--rank" else main()
