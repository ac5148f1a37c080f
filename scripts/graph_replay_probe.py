"""Diagnostic: graph replay until dfeat turns non-finite, then dissect the head (workspace, eager re-run)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition-models_amd"))
import torch
from frx import engine as E, ops
N = 256; lr = 0.005
p_ = torch.cuda.get_device_properties(0); print("uuid", getattr(p_, "uuid", None), flush=True)
def fin(t): return bool(torch.isfinite(t.float()).all())
eng = E.FaceEngine("arcface", 10575, N, dtype=ops.BF16, device="cuda:0", seed=0)
g = torch.Generator().manual_seed(1234)
batches = [((torch.rand(N, 3, 112, 112, generator=g) * 2 - 1).cuda(), torch.randint(0, 10575, (N,), generator=g).cuda()) for _ in range(4)]
images = torch.empty_like(batches[0][0]); labels = torch.empty_like(batches[0][1])
eng.net.lr_dev.fill_(lr)
def feed(i):
    images.copy_(batches[i % 4][0]); labels.copy_(batches[i % 4][1])
feed(0)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    eng.train_step(images, labels)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = eng.train_step(images, labels)
net = eng.net
for i in range(12):
    feed(i)
    pprev = net.params.clone()
    gr.replay()
    torch.cuda.synchronize()
    ok = fin(eng.dfeat)
    print(f"replay {i}: loss {out['loss'].item():.3f} feats finite {fin(net.feats)} dfeat finite {ok} params_before finite {fin(pprev)} head_w finite {fin(eng.head_w())}", flush=True)
    big = net.params.abs().max().item()
    if not ok or big > 1e3 or big != big:
        ok = False
        print("   xin finite", fin(net.xin), "images finite", fin(images), "max|param|", big, "max|param before|", pprev.abs().max().item(),
              "lr_dev", net.lr_dev.item(), "max|mom|", net.mom.abs().max().item())
        gmax = [(c.name, torch.nan_to_num(net.w_grad(c), nan=-1.0, posinf=-2.0, neginf=-2.0).abs().max().item(), fin(net.w_grad(c))) for c in net.convs]
        gmax.sort(key=lambda t: -t[1])
        print("   largest conv weight-gradient |max| (this step):", [(n, f"{v:.3g}", f) for n, v, f in gmax[:6]])
        print("   non-finite conv weight gradients:", [n for n, v, f in gmax if not f][:10])
        last = net.blocks[-1]
        gtop = net.scratch[3][:last.out.numel()].float()
        print("   top of the backward: |dfeat|", eng.dfeat.abs().max().item(), "|dfeat_t|", net.dfeat_t.float().abs().max().item(),
              "|fc_wt|", net.fc_wt.float().abs().max().item(), "|g after avgpool_bwd|", gtop.abs().max().item(),
              "|last.out|", last.out.float().abs().max().item(), "|fc dW|", net.fc_w(net.grads).abs().max().item(),
              "|top dz3|", last.dz3.float().abs().max().item())
        print("   per conv (network order) |dW|max, |dz3|/|dy2| of its block:")
        for bi, b in enumerate(net.blocks):
            row = [f"{cc.name.split('.')[-1] if 'down' not in cc.name else 'ds'}={net.w_grad(cc).abs().max().item():.2g}" for cc in (b.conv1, b.conv2, b.conv3, b.down) if cc is not None]
            print(f"     block {bi:2d}: {' '.join(row)} | dz3 {b.dz3.float().abs().max().item():.2g} dy2 {b.dy2.float().abs().max().item():.2g} "
                  f"coef(bn3/bn2/bn1) {[f'{x.abs().max().item():.2g}' for x in b.coefs[:3]]} invstd3 {net._bn(net.bn_invstd, b.conv3).max().item():.3g} invstd2 {net._bn(net.bn_invstd, b.conv2).max().item():.3g} invstd1 {net._bn(net.bn_invstd, b.conv1).max().item():.3g}")
        o = net.convs[-1].w_off + net.convs[-1].w_numel
        rest = net.grads[o:]
        print("   rest of grads (gamma/beta/fc/head): finite", fin(rest), "absmax", torch.nan_to_num(rest, nan=0.0, posinf=0.0, neginf=0.0).abs().max().item())
        d = (net.params - pprev).abs()
        print("   max |delta param| conv part", torch.nan_to_num(d[:o], nan=-1.0, posinf=-2.0).max().item(), "rest", torch.nan_to_num(d[o:], nan=-1.0, posinf=-2.0).max().item())
        for c in net.convs:
            st = dict(wk=fin(c.wk), wt=(fin(c.wt) if c.wt is not None else None), master=fin(net.w_master(c)), y=fin(c.y),
                      scale=fin(net._bn(net.bn_scale, c)), shift=fin(net._bn(net.bn_shift, c)), mean=fin(net._bn(net.bn_mean, c)),
                      invstd=fin(net._bn(net.bn_invstd, c)), gamma=fin(net.gamma(c)), beta=fin(net.beta(c)))
            if not all(st[k] for k in ("y", "scale", "shift", "mean", "invstd")):
                print("   first conv with a non-finite FORWARD quantity:", c.name, st)
                i0 = net.convs.index(c)
                if i0 > 0:
                    pc = net.convs[i0 - 1]
                    print("      previous conv", pc.name, "y absmax", pc.y.float().abs().max().item(), "scale absmax", net._bn(net.bn_scale, pc).abs().max().item(),
                          "invstd max", net._bn(net.bn_invstd, pc).max().item())
                print("      this conv: y absmax(finite part)", torch.nan_to_num(c.y.float(), nan=0.0, posinf=0.0, neginf=0.0).abs().max().item(),
                      "mean absmax", torch.nan_to_num(net._bn(net.bn_mean, c)).abs().max().item())
                y = c.y.float(); nf = ~torch.isfinite(y)
                if nf.any():
                    idx = nf.nonzero()
                    print("      y non-finite count", int(nf.sum()), "of", y.numel(), "channels", idx[:, 3].unique()[:16].tolist(), "first", idx[0].tolist())
                wk = c.wk.float(); nfw = ~torch.isfinite(wk)
                if nfw.any():
                    print("      wk non-finite count", int(nfw.sum()), "of", wk.numel(), "first", nfw.nonzero()[0].tolist(), "master there",
                          net.w_master(c).flatten()[:4].tolist())
                break
        def ru(x, m): return (x + m - 1) // m * m
        Cc, D = 10575, 512; Cpad, Npad = ru(Cc, 64), ru(N, 64)
        wsf = eng.head.ws.view(torch.float32); off = 0; regs = {}
        for nm, nfl in (("xinv", Npad), ("xnorm", Npad), ("winv", Cpad), ("ty", Npad), ("tysum", 64), ("rowloss", Npad), ("lse", Npad), ("dn", Npad),
                        ("rowrank", Npad), ("cbuf", N * Cpad), ("gbuf", N * Cpad), ("dxh", N * D), ("dwh", Cc * D)):
            regs[nm] = wsf[off:off + nfl]; off += ru(nfl * 4, 256) // 4
        print("   head workspace |max| per region:", {k: (f"{v.abs().max().item():.3g}" if k != "rowrank" else "-") for k, v in regs.items()})
        wv = regs["winv"][:Cc]; print("   winv: min", wv.min().item(), "max", wv.max().item(), "argmax", wv.argmax().item(),
                                      "| head weight row norms min", eng.head_w().view(Cc, D).norm(dim=1).min().item())
        ws = eng.head.ws.view(torch.float32)
        nf = (~torch.isfinite(ws)).nonzero().flatten()
        print("   head workspace floats:", ws.numel(), "non-finite:", nf.numel(), "first/last idx", (nf[0].item(), nf[-1].item()) if nf.numel() else None)
        print("   last dict:", {k: (fin(v) if isinstance(v, torch.Tensor) and v.dtype.is_floating_point else None) for k, v in eng.last.items()})
        # eager re-run of the head on the SAME feats / labels with the weights from before this step
        hw = pprev[-eng.head_w().numel():].view_as(eng.head_w()).contiguous()
        o2 = ops.head_forward(eng.head, net.feats, hw, labels, state_t=eng.t)
        dx2 = torch.empty_like(net.feats); dw2 = torch.empty_like(hw)
        ops.head_backward(eng.head, net.feats, hw, labels, state_t=eng.t, dx=dx2, dw=dw2)
        torch.cuda.synchronize()
        print("   eager re-run: loss", o2["loss"].item(), "dx finite", fin(dx2), "dw finite", fin(dw2), "|dx|max", dx2.abs().max().item())
        print("   labels min/max", labels.min().item(), labels.max().item(), "feats absmax", net.feats.abs().max().item())
        break
