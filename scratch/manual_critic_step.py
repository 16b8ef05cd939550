"""Prototype of the explicit critic-step gradient algorithm the HIP path implements."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from oracle import rdgan_np as onp, rdgan_torch as ot, rng as orng

def conv_wgrad(x, gy, stride, pad):
    B, Do, Ho, Wo, Cout = gy.shape
    _, D, H, W, Cin = x.shape
    need = [(o - 1) * stride + 3 for o in (Do, Ho, Wo)]
    after = [max(nd - n - p, 0) for nd, n, p in zip(need, (D, H, W), pad)]
    xp = np.pad(x, ((0,0),(pad[0],after[0]),(pad[1],after[1]),(pad[2],after[2]),(0,0)))
    dw = np.zeros((3,3,3,Cin,Cout))
    for a in range(3):
        for b in range(3):
            for c in range(3):
                patch = xp[:, a:a+(Do-1)*stride+1:stride, b:b+(Ho-1)*stride+1:stride, c:c+(Wo-1)*stride+1:stride, :]
                dw[a,b,c] = np.einsum('bdhwi,bdhwo->io', patch, gy)
    return dw

def manual_critic_grads(dpar, x_real, fake, cond, alpha, masks3):
    B = x_real.shape[0]; nd = cond.shape[1]; geo = onp.critic_geometry(nd)
    xhat = onp.random_weighted_average(x_real, fake, alpha)
    x3 = np.concatenate([x_real, fake, xhat]); c3 = np.concatenate([cond]*3)
    cin = onp.critic_input(x3, c3)
    v, inter = onp.critic_forward(dpar, x3, c3, masks3, True)
    h = [cin] + inter['h']
    gate = [np.where(inter['a'][l] > 0, 1.0, 0.2) * masks3[l] for l in range(4)]
    dv = np.concatenate([-np.ones(B)/B, np.ones(B)/B, np.ones(B)])
    w6 = dpar[8].reshape(h[4].shape[1:])
    u = [None]*5
    u[4] = gate[3] * w6[None] * dv[:,None,None,None,None]
    for l in (4,3,2):
        g = onp.conv3d_input_grad(u[l], dpar[2*(l-1)], geo[l-1][0], 2, geo[l-1][2])
        u[l-1] = gate[l-2] * g
    g0 = onp.conv3d_input_grad(u[1][2*B:], dpar[0], geo[0][0], 2, geo[0][2])[..., :1]
    n = np.sqrt((g0**2).reshape(B,-1).sum(1))
    coef = (10.0/B) * 2*(n-1)/n
    r = [None]*5
    r0 = coef[:,None,None,None,None]*g0
    inp = [h[l].copy() for l in range(5)]
    inp[0][2*B:] = np.concatenate([r0, np.zeros_like(r0)], -1)
    for l in (1,2,3,4):
        q = onp.conv3d(inp[l-1][2*B:], dpar[2*(l-1)], None, 2, geo[l-1][2], geo[l-1][1])
        inp[l][2*B:] = gate[l-1][2*B:] * q
    grads = []
    for l in (1,2,3,4):
        grads.append(conv_wgrad(inp[l-1], u[l], 2, geo[l-1][2]))
        grads.append(u[l][:2*B].sum((0,1,2,3)))
    grads.append((inp[4].reshape(3*B,-1) * dv[:,None]).sum(0)[:,None])
    grads.append(np.array([dv[:2*B].sum()]))
    losses = [np.mean(-v[:B]), np.mean(v[B:2*B]), np.mean((n-1)**2)]
    return losses, grads

rng = np.random.default_rng(5)
gpar = [p.astype(np.float64) for p in onp.init_generator(rng, 16)]
dpar = [p.astype(np.float64) for p in onp.init_critic(rng, 16)]
dpar = [p if p.ndim > 1 else 0.05*rng.standard_normal(p.shape) for p in dpar]
x, cond, z = ot.synthetic_batch(2, 16, 3, np.float64)
seed = 42; B = 2
losses, grads = ot.critic_step_grads([torch.from_numpy(p) for p in dpar], [torch.from_numpy(p) for p in gpar],
                                     torch.from_numpy(x), torch.from_numpy(cond), torch.from_numpy(z), seed)
fake = onp.generator_forward(gpar, z, cond)
alpha = orng.uniform(seed, orng.STREAM_ALPHA, B).astype(np.float64)
masks3 = [orng.dropout_scale_mask(seed, 1+i, (3*B,)+onp.critic_geometry(16)[i][1]+(c,)).astype(np.float64) for i,c in enumerate((64,128,256,256))]
ml, mg = manual_critic_grads(dpar, x, fake, cond, alpha, masks3)
print('losses', losses.numpy()[1:], ml)
for i,(a,b) in enumerate(zip(grads, mg)):
    a = a.numpy(); err = np.abs(a-b.reshape(a.shape)).max()/ (np.abs(a).max()+1e-30)
    print(i, a.shape, 'relerr', err)
