"""-m gpu: distillation -- ``losses.KlDivergence`` (mkb_kl_divergence), ``distillation.Distillation.distill``
and ``distillation.KdmkbModel.forward`` against captures of the live reference (tools/make_golden.py::gen_distill)."""
import collections

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util_gpu import grad_close  # noqa: E402  (tests/ is on sys.path: conftest.py)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_kl_divergence_vs_reference(golden, tag):
    from mkb_amd import losses

    g = golden("distill.npz")
    s = torch.tensor(g[f"kl/{tag}/student"]).cuda().requires_grad_(True)
    t = torch.tensor(g[f"kl/{tag}/teacher"]).cuda()
    loss = losses.KlDivergence()(student_score=s, teacher_score=t, T=float(g[f"kl/{tag}/T"]))
    np.testing.assert_allclose(loss.item(), float(g[f"kl/{tag}/loss"]), rtol=0, atol=1e-6)
    (2 * loss).backward()
    np.testing.assert_allclose(s.grad.cpu().numpy(), 2 * g[f"kl/{tag}/dstudent"], rtol=0, atol=1e-6)


def test_kl_divergence_teacher_gradient_matches_torch_autograd():
    from mkb_amd import losses

    torch.manual_seed(0)
    s0, t0 = torch.randn(6, 9) * 2, torch.randn(6, 9) * 2
    s, t = s0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
    ref = torch.mean(torch.nn.functional.kl_div(torch.log_softmax(s / 1.5, dim=1), torch.softmax(t / 1.5, dim=1), reduction="none"))
    ref.backward()
    sc, tc = s0.cuda().requires_grad_(True), t0.cuda().requires_grad_(True)
    loss = losses.KlDivergence()(student_score=sc, teacher_score=tc, T=1.5)
    loss.backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(sc.grad.cpu().numpy(), s.grad.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(tc.grad.cpu().numpy(), t.grad.numpy(), rtol=0, atol=1e-6)


def _model(cls, ent, rel, hidden, gamma, ents, rels):
    from mkb_amd import models

    m = getattr(models, cls)(hidden_dim=hidden, entities=ents, relations=rels, gamma=gamma)
    m._set_params(torch.as_tensor(ent), torch.as_tensor(rel))
    return m.cuda()


def test_distill_reference_doctest_known_answer(golden):
    """distillation/distillation.py:452-498: Umls, RotatE hidden 3, UniformSampling(3, 3, seed 42) -> loss 1.3066."""
    from mkb_amd import datasets, distillation

    g, gj = golden("distill.npz"), golden("distill.json")
    ds = datasets.Umls(batch_size=3, shuffle=False, seed=42, num_workers=0)
    teacher = _model("RotatE", g["umls/teacher_ent"], g["umls/teacher_rel"], 3, 6, ds.entities, ds.relations)
    student = _model("RotatE", g["umls/student_ent"], g["umls/student_rel"], 3, 6, ds.entities, ds.relations)
    proc = distillation.Distillation(teacher_entities=ds.entities, student_entities=ds.entities, teacher_relations=ds.relations,
                                     student_relations=ds.relations,
                                     sampling=distillation.UniformSampling(batch_size_entity=3, batch_size_relation=3, seed=42))
    sample = next(iter(ds))["sample"]
    np.testing.assert_array_equal(sample.numpy(), g["umls/sample"])
    loss = proc.distill(teacher=teacher, student=student, sample=sample.cuda())
    assert round(loss.item(), 4) == gj["umls_doctest_loss"] == 1.3066
    np.testing.assert_allclose(loss.item(), float(g["umls/loss"]), rtol=0, atol=1e-5)
    loss.backward()
    grad_close(student.entity_embedding.grad.cpu().numpy(), g["umls/g_ent"])
    grad_close(student.relation_embedding.grad.cpu().numpy(), g["umls/g_rel"], rtol=1e-4)
    assert teacher.entity_embedding.grad is None


def test_distill_partially_shared_graphs(golden):
    """Teacher and student know different entity / relation sets with different ids; only fully shared triples distil."""
    from mkb_amd import distillation

    g, gj = golden("distill.npz"), golden("distill.json")
    t_ents = {f"e{i}": i for i in range(6)}
    s_ents = {f"e{i}": j for j, i in enumerate([7, 3, 8, 5, 4, 6])}
    t_rels = {f"r{i}": i for i in range(3)}
    s_rels = {"r3": 0, "r1": 1, "r2": 2}
    teacher = _model("TransE", g["part/teacher_ent"], g["part/teacher_rel"], 4, 3, t_ents, t_rels)
    student = _model("DistMult", g["part/student_ent"], g["part/student_rel"], 5, 3, s_ents, s_rels)
    proc = distillation.Distillation(teacher_entities=t_ents, student_entities=s_ents, teacher_relations=t_rels,
                                     student_relations=s_rels,
                                     sampling=distillation.UniformSampling(batch_size_entity=2, batch_size_relation=2, seed=5))
    sample = torch.as_tensor(g["part/sample"])
    assert [proc.available(*row) for row in sample.tolist()] == gj["part_available"]
    loss = proc.distill(teacher=teacher, student=student, sample=sample.cuda())
    np.testing.assert_allclose(loss.item(), float(g["part/loss"]), rtol=0, atol=1e-5)
    loss.backward()
    grad_close(student.entity_embedding.grad.cpu().numpy(), g["part/g_ent"])
    grad_close(student.relation_embedding.grad.cpu().numpy(), g["part/g_rel"], rtol=1e-4)


def test_kdmkb_forward_vs_reference_capture(golden):
    """kdmkb_model.py:286-360, three steps on two CountriesS1 copies (TransE teaches RotatE and vice versa), uniform
    candidate sampler: per-step losses and the tables afterwards."""
    from mkb_amd import datasets, distillation

    g, gj = golden("distill.npz"), golden("distill.json")
    torch.manual_seed(42)
    d1 = datasets.CountriesS1(batch_size=8, seed=42, num_workers=0)
    d2 = datasets.CountriesS1(batch_size=8, seed=42, num_workers=0)
    m1 = _model("TransE", g["kd/m1_ent"], g["kd/m1_rel"], 6, 3, d1.entities, d1.relations)
    m2 = _model("RotatE", g["kd/m2_ent"], g["kd/m2_rel"], 4, 3, d2.entities, d2.relations)
    mods, dsets = collections.OrderedDict(a=m1, b=m2), collections.OrderedDict(a=d1, b=d2)
    kd = distillation.KdmkbModel(models=mods, datasets=dsets, lr={"a": 1e-2, "b": 1e-2}, alpha_kl={"a": 0.3, "b": 0.6},
                                 alpha_adv={"a": 0.5, "b": 0.5}, negative_sampling_size={"a": 4, "b": 4},
                                 batch_size_entity={"a": 5, "b": 5}, batch_size_relation={"a": 2, "b": 2},
                                 n_random_entities={"a": 3, "b": 3}, n_random_relations={"a": 1, "b": 1}, device="cuda", seed=42)
    for want in gj["kd_step_losses"]:
        kd.forward(dsets, mods, {"a": 0.3, "b": 0.6})
        got = {k: kd.metrics[k]._w[-1] for k in mods}
        assert got == pytest.approx(want, abs=2e-5), (got, want)
    for key, m in (("m1", m1), ("m2", m2)):
        np.testing.assert_allclose(m.entity_embedding.detach().cpu().numpy(), g[f"kd/{key}_ent_after"], rtol=0, atol=3e-5)
        np.testing.assert_allclose(m.relation_embedding.detach().cpu().numpy(), g[f"kd/{key}_rel_after"], rtol=0, atol=3e-5)
