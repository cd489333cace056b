"""run the full-size SpeechEnhancer step (dropout 0) several times from the same state: are the gradients reproducible?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn
from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import SpeechEnhancer, batch_stft, compute_loss
ops.set_compute_dtype(torch.float16)
Bt, Lt = 256, 64000
model = SpeechEnhancer(n_freq=129, d_model=256, num_blocks=4, num_heads=4, d_ff=1024, kernel_size=31, dropout=0.0)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, 4321).items()}
model.load_state_dict(sd)
model.cuda().train()
noisy, clean = syn.synth_wave(Bt, Lt, 777)
noisy, clean = torch.from_numpy(noisy).cuda(), torch.from_numpy(clean).cuda()
nr, ni = batch_stft(noisy, 256, 80, 160)
cr, ci = batch_stft(clean, 256, 80, 160)
ref = None
for it in range(4):
    model.load_state_dict(sd)
    model.zero_grad()
    total, _ = compute_loss(model, nr, ni, clean, cr, ci)
    total.backward()
    torch.cuda.synchronize()
    g = {k: p.grad.double().clone() for k, p in model.named_parameters()}
    gn = float(torch.sqrt(sum((v ** 2).sum() for v in g.values())))
    print("iter %d loss %.6f |g| %.5f" % (it, float(total.detach()), gn))
    if ref is None:
        ref = g
    else:
        worst = sorted(((float((g[k] - ref[k]).norm()), float(ref[k].norm()), float(g[k].norm()), k) for k in g), reverse=True)[:8]
        for d, a, b_, k in worst:
            print("   %-44s |diff| %.3e  |iter0| %.3e  |this| %.3e" % (k, d, a, b_))
