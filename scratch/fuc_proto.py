"""Prototype of the 'shared-centre' fast form of UpSampling3D(2)+Conv3D(3^3,'same') along the d axis:
out[2s] = S x[s] - A E[s],  out[2s+1] = S x[s] + D E[s+1],  E[j] = x[j]-x[j-1] (zero-extended), S = W0+W1+W2, A = W0, D = W2.
Checks forward identity and the plan-level backward (48 tap products instead of 64)."""
import numpy as np, torch, itertools
torch.manual_seed(0)
B, D, H, Wd, Ci, Co = 2, 3, 2, 2, 3, 4
x = torch.randn(B, D, H, Wd, Ci, dtype=torch.float64, requires_grad=True)
W = torch.randn(3, 3, 3, Ci, Co, dtype=torch.float64, requires_grad=True)
def direct(x, W):
    up = x.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
    y = torch.nn.functional.conv3d(up.permute(0, 4, 1, 2, 3), W.permute(4, 3, 0, 1, 2), padding=1)
    return y.permute(0, 2, 3, 4, 1)
ref = direct(x, W)
g = torch.randn_like(ref)
(ref * g).sum().backward()
dx_ref, dW_ref = x.grad.clone(), W.grad.clone()

Td = np.array([[-1, 0, 0], [1, 1, 1], [0, 0, 1]], float)           # groups A', S, D along d
Tc = np.array([[1, 0, 0], [0, 1, 1], [1, 1, 0], [0, 0, 1]], float)  # (p,t) = (0,0),(0,1),(1,0),(1,1) on a collapsed axis
coff = {(0, 0): -1, (0, 1): 0, (1, 0): 0, (1, 1): 1}                # source offset of (p,t)
Wn = W.detach().numpy(); xn = x.detach().numpy(); gn = g.numpy()
# U[g][ph,th][pw,tw]
U = np.einsum('ga,hb,wc,abcio->ghwio', Td, Tc, Tc, Wn)              # [3][4][4][Ci][Co]
xp = np.pad(xn, ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)))
E = xp[:, 1:] - xp[:, :-1]                                           # [B][D+1] (E[j] = x[j]-x[j-1]), h/w padded by 1
E = E[:, :D + 1]
out = np.zeros((B, 2 * D, 2 * H, 2 * Wd, Co))
for ph, pw in itertools.product(range(2), range(2)):
    MS = np.zeros((B, D, H, Wd, Co)); MA = np.zeros_like(MS); MD = np.zeros_like(MS)
    for th, tw in itertools.product(range(2), range(2)):
        oh, ow = coff[(ph, th)], coff[(pw, tw)]
        xs = xp[:, 1:D + 1, 1 + oh:1 + oh + H, 1 + ow:1 + ow + Wd]
        Es0 = E[:, 0:D, 1 + oh:1 + oh + H, 1 + ow:1 + ow + Wd]
        Es1 = E[:, 1:D + 1, 1 + oh:1 + oh + H, 1 + ow:1 + ow + Wd]
        MS += xs @ U[1, ph * 2 + th, pw * 2 + tw]
        MA += Es0 @ U[0, ph * 2 + th, pw * 2 + tw]
        MD += Es1 @ U[2, ph * 2 + th, pw * 2 + tw]
    out[:, 0::2, ph::2, pw::2] = MS + MA
    out[:, 1::2, ph::2, pw::2] = MS + MD
print("fwd err", np.abs(out - ref.detach().numpy()).max())

# backward, plan level
gS = gn[:, 0::2] + gn[:, 1::2]                                       # [B][D][2H][2W][Co]
dU = np.zeros_like(U)
dxS = np.zeros((B, D, H + 2, Wd + 2, Ci)); dE = np.zeros((B, D + 1, H + 2, Wd + 2, Ci))
for ph, pw, th, tw in itertools.product(range(2), repeat=4):
    oh, ow = coff[(ph, th)], coff[(pw, tw)]
    xs = xp[:, 1:D + 1, 1 + oh:1 + oh + H, 1 + ow:1 + ow + Wd]
    Es0 = E[:, 0:D, 1 + oh:1 + oh + H, 1 + ow:1 + ow + Wd]
    Es1 = E[:, 1:D + 1, 1 + oh:1 + oh + H, 1 + ow:1 + ow + Wd]
    gs = gS[:, :, ph::2, pw::2]; g0 = gn[:, 0::2, ph::2, pw::2]; g1 = gn[:, 1::2, ph::2, pw::2]
    k = (ph * 2 + th, pw * 2 + tw)
    dU[1][k] = np.einsum('bdhwi,bdhwo->io', xs, gs)
    dU[0][k] = np.einsum('bdhwi,bdhwo->io', Es0, g0)
    dU[2][k] = np.einsum('bdhwi,bdhwo->io', Es1, g1)
    dxS[:, :, 1 + oh:1 + oh + H, 1 + ow:1 + ow + Wd] += gs @ U[1][k].T
    dE[:, 0:D, 1 + oh:1 + oh + H, 1 + ow:1 + ow + Wd] += g0 @ U[0][k].T
    dE[:, 1:D + 1, 1 + oh:1 + oh + H, 1 + ow:1 + ow + Wd] += g1 @ U[2][k].T
dxS = dxS[:, :, 1:-1, 1:-1]; dE = dE[:, :, 1:-1, 1:-1]
dx = dxS + dE[:, :D] - dE[:, 1:]
dW = np.einsum('ga,hb,wc,ghwio->abcio', Td, Tc, Tc, dU)
print("dx err", np.abs(dx - dx_ref.numpy()).max(), "dW err", np.abs(dW - dW_ref.numpy()).max())
