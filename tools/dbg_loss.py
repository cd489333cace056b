import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import arr
from oracle import sfm_oracle as orc
from sincformer_metacog_speech_enhancement_amd import train, ops, functional as Fn
L = int(sys.argv[1]); B = 2
clean = arr("cw", (B, L), 70, 0.1); noisy = clean + arr("nw", (B, L), 71, 0.05)
nr, ni = orc.stft(noisy); cr, ci = orc.stft(clean)
def run(which):
    er = (nr * 0.8 + 0.05 * arr("pe", tuple(nr.shape), 72)).requires_grad_(True)
    ei = (ni * 0.8 + 0.05 * arr("pf", tuple(nr.shape), 73)).requires_grad_(True)
    enh = orc.istft(er, ei, L)
    enh.retain_grad()
    if which == "si": l = orc.si_snr_loss(enh, clean)
    elif which == "mr": l = orc.mr_stft_loss(enh, clean)
    else: l = 0.5 * (torch.sqrt(er ** 2 + ei ** 2 + 1e-8) - torch.sqrt(cr ** 2 + ci ** 2 + 1e-8)).abs().mean()
    l.backward()
    return er.grad, enh.grad
# GPU pieces
er = (nr * 0.8 + 0.05 * arr("pe", tuple(nr.shape), 72)).cuda(); ei = (ni * 0.8 + 0.05 * arr("pf", tuple(nr.shape), 73)).cuda()
cw = clean.cuda()
T = er.shape[1]
enh_wav = Fn.istft(er, ei, L)
Sw = ops.wave_moments(enh_wav, cw)
dw = torch.empty(B, L, device="cuda"); ops.sisnr_bwd(enh_wav, cw, Sw, dw)
g_er, g_w = run("si")
print("sisnr dwave err", float((dw.cpu() - g_w).abs().max()), "ref max", float(g_w.abs().max()))
inv_env = train._inv_envelope(L, T, 256, 80, 160, "cuda")
# istft adjoint alone
d_er = torch.empty_like(er); d_ei = torch.empty_like(er)
ops.framed_gemm((dw * inv_env).contiguous(), train._adjoint_consts(256, 160, "cuda")["invT"], d_er, B=B, M=T, Ls=L, sig_batch_stride=L, hop=80,
                padl=128 - 48, K=160, N=258, o_batch_stride=T * 129, ldm=129, ldn=1, mode=0, out2=d_ei, nsplit=129)
e = (d_er.cpu() - g_er).abs()
print("istft adj err max", float(e.max()), "ref max", float(g_er.abs().max()), "per-frame err", e.amax(dim=(0, 2))[-6:].tolist())
# MR part
g_er2, g_w2 = run("mr")
dw2 = torch.zeros(B, L, device="cuda")
for i, (nf, hp, wn) in enumerate(Fn.MR_STFT):
    pr, pi = Fn.stft(enh_wav, nf, hp, wn); tr, ti = Fn.stft(cw, nf, hp, wn)
    S = ops.spec_sums(pr, pi, tr, ti)
    Tr, Fr = pr.shape[1], pr.shape[2]; Mr = B * Tr; ld = ops.round_up(2 * Fr, 8)
    g = torch.zeros(Mr, ld, device="cuda")
    ops.spec_loss_bwd(pr, pi, tr, ti, S, g, g[:, Fr:], Fr, ld, 0, scale=1.0 / 3)
    frames = torch.empty(Mr, wn, device="cuda")
    ops.framed_gemm(g, train._adjoint_consts(nf, wn, "cuda")["fwdT"], frames, B=1, M=Mr, Ls=Mr * ld, sig_batch_stride=0, hop=ld, padl=0, K=2 * Fr, N=wn, o_batch_stride=0, ldm=wn, ldn=1, mode=0)
    ops.stft_adjoint_ola(frames, dw2, B, Tr, L, nf, hp, wn, accumulate=True)
e = (dw2.cpu() - g_w2).abs()
print("mr dwave err max", float(e.max()), "ref max", float(g_w2.abs().max()), "argmax", int(e.amax(0).argmax()), "tail err", e.amax(0)[-5:].tolist(), "head", e.amax(0)[:3].tolist())
