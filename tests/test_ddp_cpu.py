"""Data-parallel path (one process per GPU, gradients all-reduced by DistributedDataParallel) on CPU with the gloo
backend, world_size 2.  Kernels are shimmed (tests/cpu_shim.py); what is checked is the N>1 plumbing of PRETrainer:
batch split (IMG_PER_BATCH_UNLABEL / world), per-rank seeds, gradient averaging, identical parameters on all ranks after
a step, BatchNorm statistics staying per-rank (broadcast_buffers=False, as coin/engine/pre_train.py:59-62)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_shim import cpu_kernels
    from coin_amd.config import get_cfg
    from coin_amd.engine import PRETrainer

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(HERE, "..", "configs", "coin", "PRETRAINS", "CLIPDET_synthetic.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "AMD.COMPUTE_DTYPE", "fp32", "AMD.SYNTHETIC.HEIGHT", 96, "AMD.SYNTHETIC.WIDTH", 128,
                         "AMD.SYNTHETIC.BOXES_PER_IMAGE", 4, "SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.NUM_IMAGES", 2,
                         "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 16, "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 100, "MODEL.RPN.POST_NMS_TOPK_TRAIN", 30,
                         "AMD.TEXT_TEMPLATES", 1, "AMD.ARCH.LAYERS", [1, 1, 1, 1], "AMD.ARCH.WIDTH", 8, "AMD.ARCH.TEXT_WIDTH", 32,
                         "AMD.ARCH.TEXT_LAYERS", 2, "AMD.ARCH.TEXT_HEADS", 2, "AMD.ARCH.TEXT_DIM", 32, "AMD.ARCH.CONTEXT_LENGTH", 16,
                         "AMD.ARCH.VOCAB_SIZE", 64])
    with cpu_kernels():
        torch.manual_seed(0)  # identical initial weights on every rank (the reference broadcasts from rank 0 after loading)
        tr = PRETrainer(cfg)
        torch.manual_seed(100 + rank)
        assert tr.world_size == world and len(next(tr._data_loader_iter)[0]) == 1  # 2 images / 2 ranks
        with torch.no_grad():
            for n, p in tr.model.named_parameters():
                if n.endswith("bn3.weight"):
                    p.fill_(0.5)
        for _ in range(2):
            rec = tr.run_step()
    sd = {k: v.clone() for k, v in tr.model.state_dict().items()}
    torch.save({"sd": sd, "loss": {k: float(v) for k, v in rec.items()}}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_training_keeps_parameters_in_sync(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    a = torch.load(tmp_path / "rank0.pt")
    b = torch.load(tmp_path / "rank1.pt")
    params_equal, buffers_differ = 0, 0
    for k in a["sd"]:
        if "running_" in k or "num_batches" in k or "per_class_feat" in k or "prototype" in k:
            buffers_differ += int(not torch.equal(a["sd"][k], b["sd"][k]))
            continue
        assert torch.allclose(a["sd"][k], b["sd"][k], rtol=0, atol=1e-7), f"{k} diverged across ranks"
        params_equal += 1
    assert params_equal > 50
    assert buffers_differ > 0  # BatchNorm running statistics are per rank: different images -> different statistics
    assert a["loss"] != b["loss"]  # each rank trained on its own shard


def _coin_worker(rank, world, port, out_dir, b_ranks=None):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_shim import cpu_kernels
    from coin_amd.config import get_cfg
    from coin_amd.data.synthetic import synthetic_offline_detections
    from coin_amd.engine import CoinTrainer

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(HERE, "..", "configs", "coin", "GDINO", "foggy_synthetic.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "AMD.COMPUTE_DTYPE", "fp32", "AMD.SYNTHETIC.HEIGHT", 96, "AMD.SYNTHETIC.WIDTH", 128,
                         "AMD.SYNTHETIC.BOXES_PER_IMAGE", 6, "AMD.SYNTHETIC.NUM_IMAGES", world, "SOLVER.IMG_PER_BATCH_UNLABEL", world,
                         "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 16, "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 100, "MODEL.RPN.POST_NMS_TOPK_TRAIN", 30,
                         "MODEL.RPN.PRE_NMS_TOPK_TEST", 60, "MODEL.RPN.POST_NMS_TOPK_TEST", 20, "AMD.TEXT_TEMPLATES", 1, "MODEL.MERGE_DIM", 32,
                         "AMD.ARCH.LAYERS", [1, 1, 1, 1], "AMD.ARCH.WIDTH", 8, "AMD.ARCH.TEXT_WIDTH", 32, "AMD.ARCH.TEXT_LAYERS", 2,
                         "AMD.ARCH.TEXT_HEADS", 2, "AMD.ARCH.TEXT_DIM", 32, "AMD.ARCH.CONTEXT_LENGTH", 16, "AMD.ARCH.VOCAB_SIZE", 64,
                         "CLOUD.BURN_UP_STEP", 1, "CLOUD.PROTOTYPE_UPDATE_START", 0, "CLOUD.CLS_B_THRESH", 0.2,
                         "SOLVER.WARMUP_FACTOR", 1.0, "SOLVER.BASE_LR", 0.01])   # (a warm-up step of 1e-6 would not move an fp32 weight at all)
    with cpu_kernels():
        torch.manual_seed(0)
        tr = CoinTrainer(cfg)
        torch.manual_seed(100 + rank)
        g_det = torch.Generator().manual_seed(7 + rank)
        real_forward = tr.offline_teacher.forward

        def teacher(batched_inputs, branch=None, **kw):
            real_forward(batched_inputs, branch=branch, **kw)
            # b_ranks: only these ranks' teachers disagree with the cloud detector about labels (-> inconsistent 'B' boxes)
            frac = 0.25 if b_ranks is None else (1.0 if rank in b_ranks else 0.0)
            extra = 8 if b_ranks is None or rank in b_ranks else 0   # (a private box that overlaps a cloud box of another class is a B box too)
            return [synthetic_offline_detections(tr.model_CLOUD.entry(d["file_name"]), g_det, relabel_frac=frac, extra=extra) for d in batched_inputs]

        tr.offline_teacher.forward = teacher
        with torch.no_grad():
            for n, p in tr.model.named_parameters():
                if n.endswith("bn3.weight"):
                    p.fill_(0.5)
        merge0 = {k: v.clone() for k, v in tr.merge.state_dict().items()}
        had_merge, ema_due = [], []
        for _ in range(2):  # step_one, then step_two with the EMA teacher update
            ema_due.append(tr._ema_due(tr.iter))
            rec = tr.run_step()
            had_merge.append("loss_merge_a" in rec)
    torch.save({"sd": {k: v.clone() for k, v in tr.model.state_dict().items()}, "merge": {k: v.clone() for k, v in tr.merge.state_dict().items()},
                "merge0": merge0, "had_merge": had_merge, "ema_due": ema_due,
                "teacher": {k: v.clone() for k, v in tr.offline_teacher.state_dict().items()},
                "loss": {k: float(v) for k, v in rec.items()}}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_cointrainer_keeps_student_and_ckg_in_sync(tmp_path):
    """targetDET step under DDP (trainer.py:66-72: student AND merge module are wrapped): after a step_one and a step_two step the
    student's and the CKG module's parameters are identical on both ranks although each rank saw its own images."""
    world, port = 2, _free_port()
    mp.start_processes(_coin_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    a, b = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    n = 0
    for k in a["sd"]:
        if "running_" in k or "num_batches" in k or "per_class_feat" in k or "prototype" in k:
            continue
        assert torch.allclose(a["sd"][k], b["sd"][k], rtol=0, atol=1e-7), f"student {k} diverged across ranks"
        n += 1
    assert n > 50
    for k in a["merge"]:
        assert torch.allclose(a["merge"][k], b["merge"][k], rtol=0, atol=1e-7), f"CKG {k} diverged across ranks"
    assert a["loss"] != b["loss"]
    # the CKG update ran (at least one rank had B boxes; a rank without them joins the all-reduce with a zero gradient)
    assert "loss_merge_grad" in a["loss"] or "loss_merge_grad" in b["loss"]


def test_four_rank_gloo_cointrainer_b_boxes_on_a_strict_subset_of_ranks(tmp_path):
    """Four ranks, an EMA-due step_two iteration, and inconsistent ('B') boxes on ranks 0 and 1 ONLY: the CKG module's gradient
    all-reduce is entered by all four ranks (ranks 2 and 3 contribute zeros -- the reference, trainer.py:66-72 + 192-197, would
    dead-lock here), the CKG parameters move and stay identical everywhere, the student stays identical, and every rank's EMA
    teacher follows the (identical) student."""
    world, port = 4, _free_port()
    mp.start_processes(_coin_worker, args=(world, port, str(tmp_path), (0, 1)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"rank{i}.pt") for i in range(world)]
    assert all(x["ema_due"] == [False, True] for x in r)
    with_b = [i for i in range(world) if any(r[i]["had_merge"])]
    assert with_b and set(with_b) <= {0, 1}, with_b            # a strict, non-empty subset of the ranks saw B boxes
    for i in range(1, world):
        for k in r[0]["sd"]:
            if "running_" in k or "num_batches" in k or "per_class_feat" in k or "prototype" in k:
                continue
            assert torch.allclose(r[0]["sd"][k], r[i]["sd"][k], rtol=0, atol=1e-7), f"student {k} diverged on rank {i}"
        for k in r[0]["merge"]:
            assert torch.allclose(r[0]["merge"][k], r[i]["merge"][k], rtol=0, atol=1e-7), f"CKG {k} diverged on rank {i}"
    assert any(not torch.equal(r[0]["merge"][k], r[0]["merge0"][k]) for k in r[0]["merge"]), "the CKG update did not run"
    # EMA: the teachers' float parameters agree across ranks (per-rank buffers aside), since the students do
    n = 0
    for k, v in r[0]["teacher"].items():
        if v.dtype != torch.float32 or "running_" in k or "per_class_feat" in k or "prototype" in k:
            continue
        assert torch.allclose(v, r[3]["teacher"][k], rtol=0, atol=1e-6), f"teacher {k} differs between ranks 0 and 3"
        n += 1
    assert n > 50


def _reducer_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coin_amd.parallel import GradReducer, broadcast_parameters

    def make():
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1), torch.nn.Flatten(),
                                torch.nn.Linear(4 * 6 * 6, 5), torch.nn.Linear(5, 2))
        m[0].weight.data = m[0].weight.data.contiguous(memory_format=torch.channels_last)  # a non-default dense layout
        return m

    a, b = make(), make()                       # a: reducer, b: plain autograd (reference of the local gradients)
    with torch.no_grad():
        for p in a.parameters():
            p.add_(rank)                        # replicas start different ...
    broadcast_parameters(a)                     # ... and are synchronised from rank 0 once
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    red = GradReducer(list(a.parameters()), slice_mb=0.0005)   # ~130 floats per slice: several slices, launched in order
    assert len(red.slices) >= 4
    ok = True
    for step in range(3):
        g = torch.Generator().manual_seed(10 * step + rank)
        x = torch.randn(2, 3, 6, 6, generator=g)
        for m in (a, b):
            for p in m.parameters():
                p.grad = None
        skip_last = step == 1 and rank == 1      # this rank's last layer gets no gradient in step 1: zeros are contributed
        for m in (a, b):
            h = m[:5](x)
            (h.sum() if skip_last else m[5](h).square().sum()).backward()
        scale = red.finalize()
        assert scale == 1.0 / world
        for pa, pb in zip(a.parameters(), b.parameters()):
            local = pb.grad if pb.grad is not None else torch.zeros_like(pb)
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local.contiguous())
            want = sum(gathered) / world
            ok &= torch.allclose(pa.grad * scale, want, rtol=1e-6, atol=1e-7) and pa.grad.stride() == pa.stride()
    torch.save({"ok": bool(ok)}, os.path.join(out_dir, f"red{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_averages_in_slices_with_missing_gradients(tmp_path):
    """coin_amd.parallel.GradReducer on 2 gloo ranks: the reduced gradient is the mean of the ranks' local gradients, slices are
    launched in the same order on every rank even when one rank has no gradient for some parameters, layouts are preserved."""
    world, port = 2, _free_port()
    mp.start_processes(_reducer_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    assert all(torch.load(tmp_path / f"red{r}.pt")["ok"] for r in range(world))


def _collect_worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coin_amd.engine.collect import collect_clip_results
    from coin_amd.structures import Boxes, Instances

    class Relabel(torch.nn.Module):  # stands in for the CLIP teacher: (inputs, cached cloud result) -> relabelled result
        def forward(self, inputs, pre):
            inst = Instances((4, 4))
            inst.pred_boxes = Boxes(torch.tensor([[0.0, 0.0, 1.0 + rank, 2.0]]))
            return {"file_name": inputs[0]["file_name"], "image_id": inputs[0]["image_id"], "height": 4, "width": 4, "rank": rank,
                    "RCNN": {"instances": inst}, "RPN": {"instances": inst}}

    names = [f"img{i}.png" for i in range(5)]
    mine = [{"file_name": n, "image_id": n} for n in names[rank::world]]     # each rank's shard of the loader
    res = collect_clip_results(Relabel(), mine, lambda fn: {"file_name": fn}, dataset_name="d")
    got = res.get_results()["d"]
    ok = sorted(got) == names and all(got[n]["rank"] == i % world for i, n in enumerate(names))
    torch.save({"ok": bool(ok)}, os.path.join(out_dir, f"col{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_collection_loop_unions_the_ranks_shards(tmp_path):
    """CLIP_COLLECTOR.collect (clip_collector.py:46-63): every rank relabels its shard, all ranks end with the union."""
    world, port = 2, _free_port()
    mp.start_processes(_collect_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    assert all(torch.load(tmp_path / f"col{r}.pt")["ok"] for r in range(world))


def test_grad_reducer_refuses_a_contribution_that_arrives_after_an_early_delivery_was_reduced():
    """Round 6 (found with two ranks on one GPU, tests/test_ddp_gpu.py): a replayed step graph hands a parameter's gradient to the reducer
    EARLY (`deliver_early`); if another differentiable pass of the same step also feeds that parameter, the autograd engine adds its share
    in place AFTER the slice was all-reduced -- per rank, on top of the reduced sum.  The reducer notices when the engine's hook for that
    parameter arrives (the arena's version counter moved since the collective was launched) and raises; a contribution that lands while
    the slice is still open is simply part of what gets packed.  One process, no process group (the collective is skipped at world size
    1, the bookkeeping is the same)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from coin_amd.parallel import GradReducer

    a = torch.nn.Linear(4, 3)
    b = torch.nn.Linear(3, 2)
    params = list(a.parameters()) + list(b.parameters())
    red = GradReducer(params, slice_mb=0.00005)   # ~13 floats per slice; slices follow the expected arrival order (last layer first)
    assert len(red.slices) >= 2
    first = list(red.slices[0].params)             # launched first: collectives leave in slice order
    w = first[0]
    # (1) late contribution AFTER the launch: refused
    for p in first:
        p.grad = torch.ones_like(p)
        red.deliver_early(p)                       # the replay's share is in place; the last one completes the slice: launched at once
    s = red.slices[0]
    assert s.launched and w.grad.data_ptr() == s.views[0].data_ptr()
    w.grad.add_(1.0)                               # what the engine's accumulator does with the other pass's share
    with pytest.raises(RuntimeError, match="delivered early"):
        red._on_grad(w)                            # the engine's post-accumulate hook for w
    red.finalize()
    # (2) no late contribution: the engine's hook for an early parameter is swallowed silently
    for p in params:
        p.grad = None
    for p in first:
        p.grad = torch.ones_like(p)
        red.deliver_early(p)
    for p in first:
        red._on_grad(p)
    for p in params:
        if all(p is not q for q in first):
            p.grad = torch.zeros_like(p)
            red._on_grad(p)
    assert red.finalize() == 1.0
    assert torch.equal(w.grad, torch.ones_like(w))
    red.remove()
