"""SURVEY 8f N4 on the GPU: MetacognitiveArbitrationAgent (agents/maa.py) and VectorQuantizer (models/vq.py) mirrors vs the
reference's golden outputs / gradients (tests/golden/g13_routing.npz) and vs the oracle on larger seeded inputs."""
import numpy as np
import pytest
import torch

from helpers import gold, arr, maxerr, rmse, STATE_TABLES
from oracle import sfm_oracle as orc
from sincformer_metacog_speech_enhancement_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]


def _maa(seed=130):
    from sincformer_metacog_speech_enhancement_amd.agents import MetacognitiveArbitrationAgent
    m = MetacognitiveArbitrationAgent()
    ref = STATE_TABLES["MetacognitiveArbitrationAgent"]["state"]
    sd = m.state_dict()
    assert set(sd) == set(ref) and all(list(sd[k].shape) == ref[k][0] for k in sd)
    new = {}
    for k, v in sd.items():
        if k.startswith("decision_net"):
            new[k] = torch.from_numpy(syn.synth_array("maa." + k, tuple(v.shape), seed, 0.6 if k.endswith("weight") else 0.3))
        else:
            new[k] = v.clone()
    new["running_mean"], new["running_var"] = torch.tensor(0.8), torch.tensor(0.09)
    m.load_state_dict(new, strict=True)
    return m.cuda(), new


def test_maa_eval_matches_the_reference():
    g = gold("g13_routing")
    m, _ = _maa()
    m.eval()
    with torch.no_grad():
        r = m(torch.from_numpy(g["sigma"]).cuda())
    for k in ("probs", "logits", "confidence"):
        assert maxerr(r[k].cpu(), g["eval." + k]) < 3e-5, k
    assert np.array_equal(r["decisions"].cpu().numpy(), g["eval.decisions"])
    assert r["decisions"].dtype == torch.int64 and float(r["threshold"]) == 0.5
    assert m.get_strategy_name(2).startswith("HARD_MASK") and m.count_parameters() == STATE_TABLES["MetacognitiveArbitrationAgent"]["params"]


@pytest.mark.parametrize("dt", DTYPES)
def test_maa_train_mode_matches_the_reference(dt):
    """train(): EMA of the batch statistics (unbiased variance), forward on the UPDATED statistics, and the gradients of
    sigma and of every parameter for fixed cotangents on logits / probs / confidence / threshold"""
    from sincformer_metacog_speech_enhancement_amd import ops
    ops.set_compute_dtype(dt)
    g = gold("g13_routing")
    m, _ = _maa()
    m.train()
    sg = torch.from_numpy(g["sigma"]).cuda().requires_grad_(True)
    r = m(sg)
    cl, cp_, cc = (torch.from_numpy(syn.synth_array(n, tuple(r[k].shape), s)).cuda()
                   for n, k, s in (("g13_cl", "logits", 132), ("g13_cp", "probs", 133), ("g13_cc", "confidence", 134)))
    ((r["logits"] * cl).sum() + (r["probs"] * cp_).sum() + (r["confidence"] * cc).sum() + 3.0 * r["threshold"].sum()).backward()
    for k in ("probs", "logits", "confidence"):
        assert maxerr(r[k].detach().cpu(), g["train." + k]) < 3e-5, k
    assert abs(float(m.running_mean) - float(g["train_running_mean"])) < 1e-6
    assert abs(float(m.running_var) - float(g["train_running_var"])) < 1e-6
    assert int(m.num_updates) == 1
    assert maxerr(sg.grad.cpu(), g["train_dsigma"]) < 1e-4 * float(np.abs(g["train_dsigma"]).max())
    tol = 4e-3 if dt is torch.float16 else 3e-2                 # weight gradients: 16-bit GEMM operands
    for k, p_ in m.named_parameters():
        e = g["train.grad." + k]
        err = rmse(p_.grad.cpu(), e) / (float(np.sqrt((e ** 2).mean())) + 1e-12)
        assert err < (1e-6 if k == "threshold" else tol), (k, err)


@pytest.mark.parametrize("B,T", [(1, 1), (5, 1037), (64, 4000)])
def test_maa_vs_oracle_sizes(B, T):
    m, sd = _maa(seed=140)
    m.eval()
    sigma = (arr("maa_sig", (B, 1, T), 141).abs() + 0.3)
    with torch.no_grad():
        r = m(sigma.cuda())
    o, _ = orc.maa_forward(sd, sigma)
    assert maxerr(r["logits"].cpu(), o["logits"]) < 5e-5 * max(1.0, float(o["logits"].abs().max()))
    assert maxerr(r["probs"].cpu(), o["probs"]) < 3e-5 and maxerr(r["confidence"].cpu(), o["confidence"]) < 3e-6
    margin = o["logits"].topk(2, dim=-1)[0]
    clear = (margin[..., 0] - margin[..., 1]) > 1e-3             # argmax is only defined up to rounding at ties
    assert torch.equal(r["decisions"].cpu()[clear], o["decisions"][clear])


def test_vector_quantizer_matches_the_reference():
    from sincformer_metacog_speech_enhancement_amd.models.vq import VectorQuantizer
    g = gold("g13_routing")
    vq = VectorQuantizer().cuda()
    assert list(vq.state_dict()) == list(STATE_TABLES["VectorQuantizer"]["state"])
    assert maxerr(vq.centroids.detach().cpu(), torch.linspace(0, 1, 3)) == 0
    with torch.no_grad():
        vq.centroids.copy_(torch.tensor([0.07, 0.46, 0.93]))
    x = torch.from_numpy(g["vq_x"]).cuda().requires_grad_(True)
    q, idx, loss = vq(x)
    cq = torch.from_numpy(syn.synth_array("g13_cq", tuple(q.shape), 136)).cuda()
    ((q * cq).sum() + 1.7 * loss).backward()
    assert maxerr(q.detach().cpu(), g["vq_q"]) == 0 and np.array_equal(idx.cpu().numpy(), g["vq_idx"])
    assert idx.dtype == torch.int64 and abs(float(loss) - float(g["vq_loss"])) < 1e-7
    assert maxerr(x.grad.cpu(), g["vq_dx"]) < 1e-6 and maxerr(vq.centroids.grad.cpu(), g["vq_dcentroids"]) < 1e-6
    assert maxerr(vq.get_utilization(idx), g["vq_utilization"]) < 1e-6 and maxerr(vq.get_centroids().detach().cpu(), g["vq_sorted"]) == 0


@pytest.mark.parametrize("shape,M", [((1,), 3), ((4, 801, 129), 3), ((3, 257), 16), ((2, 5, 7), 1)])
def test_vector_quantizer_vs_oracle_and_properties(shape, M):
    """values, indices, loss and gradients vs the oracle; quantisation is idempotent (q(q(x)) = q(x), loss 0)"""
    from sincformer_metacog_speech_enhancement_amd.models.vq import VectorQuantizer, VQMaskQuantizer
    vq = VectorQuantizer(num_centroids=M).cuda()
    with torch.no_grad():
        vq.centroids.add_(0.03 * arr("vq_c", (M,), 150).cuda())
    x = (0.5 + 0.4 * arr("vq_xx", shape, 151)).requires_grad_(True)
    xg = x.detach().cuda().requires_grad_(True)
    q, idx, loss = vq(xg)
    cen = vq.centroids.detach().cpu().clone().requires_grad_(True)
    qo, io, lo = orc.vq_forward(cen, x)
    assert maxerr(q.detach().cpu(), qo.detach()) == 0 and torch.equal(idx.cpu(), io)
    assert abs(float(loss) - float(lo)) < 1e-6 * max(1.0, float(lo))
    (q.sum() * 0.5 + loss).backward()
    (qo.sum() * 0.5 + lo).backward()
    assert maxerr(xg.grad.cpu(), x.grad) < 1e-6
    assert maxerr(vq.centroids.grad.cpu(), cen.grad) < 1e-5 * max(1.0, float(cen.grad.abs().max()))
    with torch.no_grad():
        q2, idx2, loss2 = vq(q.detach())
    # (the reference returns x + (c - x), which can sit one ulp off the centroid c)
    assert maxerr(q2.cpu(), q.detach().cpu()) < 2e-7 and torch.equal(idx2, idx) and float(loss2) < 1e-12
    wrapped = VQMaskQuantizer(torch.nn.Sigmoid(), num_centroids=3).cuda()
    qm, sm, vl = wrapped(xg.detach(), return_soft=True)
    assert qm.shape == sm.shape == xg.shape and vl.dim() == 0


def test_routing_modules_refuse_cpu_tensors():
    from sincformer_metacog_speech_enhancement_amd.agents import MetacognitiveArbitrationAgent
    from sincformer_metacog_speech_enhancement_amd.models.vq import VectorQuantizer
    with pytest.raises(RuntimeError):
        MetacognitiveArbitrationAgent()(torch.rand(2, 5))
    with pytest.raises(RuntimeError):
        VectorQuantizer()(torch.rand(2, 5))
