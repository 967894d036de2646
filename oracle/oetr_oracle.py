"""CPU oracle for the OETR feature-correlation + overlap-regression path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``imagematching_oetr_amd/`` may
import this file; it is used by ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` as the *checker* / the timed CPU
baseline, never as the thing shipped.

It is a functional restatement (plain tensors + a flat ``{name: tensor}``
weight dict keyed like a reference checkpoint) of the algorithm in the
reference's Python sources; every function cites the lines it follows.  The
arithmetic primitives are torch's CPU kernels - the same third-party
dependency (torch, version pinned by this image: 2.10.0) the reference itself
dispatches to, so `torch` IS the numeric ground truth here.  Parity is pinned
by ``tests/golden/*.npz``: outputs of the imported reference
(``oracle/gen_golden.py``, run in the build container where /root/reference
exists) which ``tests/test_oracle_golden.py`` replays against this file.
Known-answer vectors from the reference: the ``bbox_overlaps`` docstring
(reference ``src/losses/utils.py:31-53``) - the only ones the repo holds.
Parity on TRAINED weights is unpinned: no checkpoint is reachable offline
(SURVEY.md §8c); all vectors use seeded synthetic weights.

``dtype=torch.float64`` runs the same graph in double precision to measure
how far fp32 implementations may legitimately drift (tolerances in tests/).
"""
import math

import torch
import torch.nn.functional as F

D_MODEL = 256
N_HEAD = 8
HEAD_DIM = D_MODEL // N_HEAD
N_ENC = 8          # self,cross,self,cross,... (reference transformer.py:295)
N_DEC = 2          # reference transformer.py:304
LN_EPS = 1e-5
ATTN_EPS = 1e-6    # reference linear_attention.py:17
GN_GROUPS = 32     # reference model.py:74


# --------------------------------------------------------------------------
# synthetic weights (shared by the golden generator, tests and bench)
# --------------------------------------------------------------------------
def hot_path_param_shapes():
    """name -> shape for every tensor the hot path reads, using the key names
    of a reference ``OETR.state_dict()``."""
    C = D_MODEL
    s = {}
    for i in range(N_ENC):
        p = f'transformer.encoder.{i}.'
        for n in ('q_proj', 'k_proj', 'v_proj', 'merge'):
            s[p + n + '.weight'] = (C, C)
        s[p + 'mlp.0.weight'] = (2 * C, C)
        s[p + 'mlp.2.weight'] = (C, 2 * C)
        for n in ('pre_norm_q', 'pre_norm_kv', 'norm2'):
            s[p + n + '.weight'] = (C,)
            s[p + n + '.bias'] = (C,)
    for i in range(N_DEC):
        p = f'transformer.decoder.layers.{i}.'
        for a in ('self_attn', 'multihead_attn'):
            for n in ('q_proj', 'k_proj', 'v_proj'):
                s[p + f'{a}.{n}.weight'] = (C, C)
                s[p + f'{a}.{n}.bias'] = (C,)
            s[p + f'{a}.merge.weight'] = (C, C)
        s[p + 'mlp.0.weight'] = (2 * C, C)
        s[p + 'mlp.2.weight'] = (C, 2 * C)
        for n in ('norm1', 'norm2', 'norm3'):
            s[p + n + '.weight'] = (C,)
            s[p + n + '.bias'] = (C,)
    s['query_embed1.weight'] = (1, C)
    s['query_embed2.weight'] = (1, C)
    s['tlbr_reg.0.weight'] = (C, C)
    s['tlbr_reg.2.weight'] = (4, C)
    s['tlbr_reg.2.bias'] = (4,)
    s['heatmap_conv.0.weight'] = (C, C, 3, 3)
    s['heatmap_conv.0.bias'] = (C,)
    s['heatmap_conv.1.weight'] = (C,)
    s['heatmap_conv.1.bias'] = (C,)
    s['heatmap_conv.3.weight'] = (1, C, 1, 1)
    s['heatmap_conv.3.bias'] = (1,)
    return s


def make_hot_weights(seed, sharpen=False):
    """Deterministic synthetic hot-path weights built from ``torch.rand`` only
    (uniform draws from the CPU mt19937 stream are bit-reproducible across
    machines; normal draws go through vectorised libm and are not relied on).

    Matrices get Xavier-uniform bounds (what the reference applies to the
    transformer, transformer.py:308-311); norm scales are 1 +- 0.2 and every
    bias is +-0.1 so that a kernel which drops a scale/bias is caught.
    ``sharpen`` rescales the two head output layers (heat-map 1x1 conv x8,
    tlbr output x0.35) so that soft-argmax is moderately peaked and the
    sigmoid stays off saturation - the regime where boxes actually depend on
    the inputs (random-init boxes are nearly input-blind)."""
    g = torch.Generator().manual_seed(int(seed))
    w = {}
    for name, shape in hot_path_param_shapes().items():
        u = torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1
        if len(shape) >= 2 and 'query_embed' not in name:
            fan_out = shape[0] * (shape[2] * shape[3] if len(shape) == 4 else 1)
            fan_in = shape[1] * (shape[2] * shape[3] if len(shape) == 4 else 1)
            w[name] = u * math.sqrt(6.0 / (fan_in + fan_out))
        elif 'query_embed' in name:
            w[name] = u * 1.7          # ~unit variance like nn.Embedding
        elif name.endswith('.weight'):  # LayerNorm / GroupNorm scale
            w[name] = 1.0 + 0.2 * u
        else:
            w[name] = 0.1 * u
    if sharpen:
        w['heatmap_conv.3.weight'] = w['heatmap_conv.3.weight'] * 8.0
        w['tlbr_reg.2.weight'] = w['tlbr_reg.2.weight'] * 0.35
    return w


def make_features(seed, n, hf, wf, scale=1.0):
    """Synthetic backbone output [n,256,hf,wf]: uniform, std ~0.29*scale, the
    spread the real (random-init) extraction path shows."""
    g = torch.Generator().manual_seed(int(seed))
    return (torch.rand(n, D_MODEL, hf, wf, generator=g) - 0.5) * scale


def make_masks(seed, n, hf, wf, kind='pad'):
    """Synthetic forward_dummy masks [n,hf,wf] (float 0/1) like a padded batch's
    (``resize_mask``, src/model.py:256-258): kind 'pad' = image i's valid region is
    its top-left hv x wv corner (hv, wv drawn per image, at least half of the grid,
    image 0 fully valid); 'holes' = 'pad' with ~10 % of the valid tokens cleared as
    well (masks need not be rectangles: the reference only multiplies by them)."""
    g = torch.Generator().manual_seed(int(seed))
    m = torch.zeros(n, hf, wf)
    for i in range(n):
        hv = hf if i == 0 else int(torch.randint((hf + 1) // 2, hf + 1, (1,), generator=g))
        wv = wf if i == 0 else int(torch.randint((wf + 1) // 2, wf + 1, (1,), generator=g))
        m[i, :hv, :wv] = 1.0
    if kind == 'holes':
        m = m * (torch.rand(n, hf, wf, generator=g) >= 0.1).float()
    return m


def checksum(t):
    """Exact, machine-independent fingerprint of an fp32 tensor: integer sums
    over the IEEE bit patterns (int64 wrap-around arithmetic is associative,
    so thread count / reduction order cannot change it, unlike a float sum).
    Used to verify that seeded tensors regenerate bit-identically elsewhere."""
    bits = t.detach().to(torch.float32).contiguous().flatten().view(torch.int32)
    bits = bits.to(torch.int64)
    idx = torch.arange(1, bits.numel() + 1, dtype=torch.int64)
    return [int(bits.sum()), int((bits * (idx % 97)).sum()),
            int((bits * (idx % 8191)).sum())]


# --------------------------------------------------------------------------
# neck: input_proj -> PatchMerging -> input_proj2   
# --------------------------------------------------------------------------
BACKBONE_C = 1024            # ResNet-50 layer3 channels (reference default.py LAST_LAYER)
PATCH_SIZES = (4, 8, 16)     # reference model.py:51-55


def neck_param_shapes():
    """name -> shape of the neck tensors, reference state-dict key names
    (src/model.py:44-56, src/models/backbone.py:27-51)."""
    C = D_MODEL
    s = {'input_proj.weight': (C, BACKBONE_C, 1, 1), 'input_proj.bias': (C,),
         'patchmerging.norm.weight': (C,), 'patchmerging.norm.bias': (C,)}
    for i, ps in enumerate(PATCH_SIZES):
        # backbone.py:38-42: 2C/2^(i+1) channels, the last one 2C/2^i
        out = 2 * C // 2 ** i if i == len(PATCH_SIZES) - 1 else 2 * C // 2 ** (i + 1)
        s[f'patchmerging.reductions.{i}.weight'] = (out, C, ps, ps)
        s[f'patchmerging.reductions.{i}.bias'] = (out,)
    s['input_proj2.weight'] = (C, 2 * C, 1, 1)
    s['input_proj2.bias'] = (C,)
    return s


def make_neck_weights(seed):
    """Seeded synthetic neck weights (same recipe as ``make_hot_weights``:
    ``torch.rand`` only; conv matrices Xavier-uniform, norm scale 1 +- 0.2,
    biases +-0.1)."""
    g = torch.Generator().manual_seed(int(seed))
    w = {}
    for name, shape in neck_param_shapes().items():
        u = torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1
        if len(shape) == 4:
            rf = shape[2] * shape[3]
            w[name] = u * math.sqrt(6.0 / ((shape[0] + shape[1]) * rf))
        elif name.endswith('norm.weight'):
            w[name] = 1.0 + 0.2 * u
        else:
            w[name] = 0.1 * u
    return w


def make_backbone_features(seed, n, hb, wb):
    """Synthetic ResNet layer3 output [n,1024,hb,wb]: non-negative (post-ReLU),
    about half the entries zero."""
    g = torch.Generator().manual_seed(int(seed))
    u = torch.rand(n, BACKBONE_C, hb, wb, generator=g)
    return torch.clamp(u * 2 - 1, min=0)


def neck(bb, w, return_stages=False):
    """Backbone features [n,1024,hb,wb] -> feat [n,256,hb//2,wb//2].

    reference src/model.py:113-118 (``input_proj2(patchmerging(input_proj(.)))``)
    with PatchMerging.forward, backbone.py:53-67: LayerNorm over channels at
    every position, three stride-2 convs (kernel 4/8/16, padding (k-2)/2) whose
    outputs are concatenated along channels (256+128+128)."""
    x = F.conv2d(bb, w['input_proj.weight'], w['input_proj.bias'])
    n, c, h, ww = x.shape
    t = x.flatten(2).transpose(1, 2)                       # n (h w) c
    t = F.layer_norm(t, (c,), w['patchmerging.norm.weight'],
                     w['patchmerging.norm.bias'], 1e-5)
    xn = t.transpose(1, 2).reshape(n, c, h, ww).contiguous()
    ys = []
    for i, ps in enumerate(PATCH_SIZES):
        ys.append(F.conv2d(xn, w[f'patchmerging.reductions.{i}.weight'],
                           w[f'patchmerging.reductions.{i}.bias'], stride=2,
                           padding=(ps - 2) // 2))
    y = torch.cat(ys, dim=1)
    feat = F.conv2d(y, w['input_proj2.weight'], w['input_proj2.bias'])
    if return_stages:
        return dict(proj=x, xn=xn, merged=y, feat=feat)
    return feat


# --------------------------------------------------------------------------
# position table (reference src/models/utils.py:174-205)
# --------------------------------------------------------------------------
def position_table(hf, wf, d_model=D_MODEL, dtype=torch.float32):
    """[1,d_model,hf,wf] window of the sine table, with the reference's
    operator-precedence quirk: ``-math.log(10000.0) / d_model // 2`` is
    ``floor((-ln1e4/d_model)/2)`` (= -1.0 at d_model=256), utils.py:188-190.
    Positions are 1-based (``ones.cumsum``, :186-187); channel c%4 selects
    sin x, cos x, sin y, cos y (:192-195).  Always evaluated in fp32 like the
    reference buffer, then cast."""
    slope = (-math.log(10000.0) / d_model) // 2
    freq = torch.exp(torch.arange(0, d_model // 2, 2).float() * slope)
    freq = freq.view(-1, 1, 1)
    yy = torch.arange(1, hf + 1).float().view(1, hf, 1).expand(1, hf, wf)
    xx = torch.arange(1, wf + 1).float().view(1, 1, wf).expand(1, hf, wf)
    pe = torch.empty(d_model, hf, wf)
    pe[0::4] = torch.sin(xx * freq)
    pe[1::4] = torch.cos(xx * freq)
    pe[2::4] = torch.sin(yy * freq)
    pe[3::4] = torch.cos(yy * freq)
    return pe.unsqueeze(0).to(dtype)


# --------------------------------------------------------------------------
# attention kernels (reference src/models/linear_attention.py)
# --------------------------------------------------------------------------
def linear_attention(q, k, v, eps=ATTN_EPS, q_mask=None, kv_mask=None):
    """q [N,L,H,D], k,v [N,S,H,D] -> [N,L,H,D].  linear_attention.py:22-50:
    phi = elu+1 (:12-13), the optional masks q_mask [N,L] / kv_mask [N,S]
    MULTIPLY phi(q) / phi(k) and v (:37-41 - "set padded position to zero"),
    v/S with S the full source length (:43-44), KV = sum_s phi(k) v (:45),
    Z = 1/(phi(q).sum_s phi(k) + eps) (:46), out = phi(q) KV Z * S (:47-48)."""
    S = v.shape[1]
    fq = F.elu(q) + 1
    fk = F.elu(k) + 1
    if q_mask is not None:
        fq = fq * q_mask[:, :, None, None]
    if kv_mask is not None:
        fk = fk * kv_mask[:, :, None, None]
        v = v * kv_mask[:, :, None, None]
    vs = v / S
    kv = torch.einsum('nshd,nshv->nhdv', fk, vs)
    z = 1 / (torch.einsum('nlhd,nhd->nlh', fq, fk.sum(dim=1)) + eps)
    return (torch.einsum('nlhd,nhdv,nlh->nlhv', fq, kv, z) * S).contiguous()


def full_attention(q, k, v):
    """Softmax attention variant, linear_attention.py:53-87 (no dropout, no
    mask): softmax over s of q.k/sqrt(D) (:73,80-81) then A.v (:85)."""
    qk = torch.einsum('nlhd,nshd->nlsh', q, k)
    a = torch.softmax(qk / q.shape[3] ** 0.5, dim=2)
    return torch.einsum('nlsh,nshd->nlhd', a, v).contiguous()


def _ln(x, w, prefix):
    return F.layer_norm(x, (x.shape[-1],), w[prefix + '.weight'],
                        w[prefix + '.bias'], LN_EPS)


def _heads(t):
    return t.view(t.shape[0], t.shape[1], N_HEAD, HEAD_DIM)


# --------------------------------------------------------------------------
# encoder / decoder (reference src/models/transformer.py)
# --------------------------------------------------------------------------
def encoder_layer(x, src, x_pos, s_pos, w, p, attention=linear_attention,
                  x_mask=None, source_mask=None):
    """transformer.py:104-142.  q = LN_q(x)+x_pos; k = v = LN_kv(src)+s_pos
    (V also receives the position term, :123-126); bias-free projections;
    attention(q_mask=x_mask, kv_mask=source_mask) (:131-136); merge;
    x += msg; x += W2 gelu_erf(W1 LN2(x))."""
    q = _ln(x, w, p + 'pre_norm_q') + x_pos
    kv = _ln(src, w, p + 'pre_norm_kv') + s_pos
    Q = _heads(F.linear(q, w[p + 'q_proj.weight']))
    K = _heads(F.linear(kv, w[p + 'k_proj.weight']))
    V = _heads(F.linear(kv, w[p + 'v_proj.weight']))
    if x_mask is None and source_mask is None:
        msg = attention(Q, K, V).reshape(x.shape)
    else:
        msg = attention(Q, K, V, q_mask=x_mask, kv_mask=source_mask).reshape(x.shape)
    x = x + F.linear(msg, w[p + 'merge.weight'])
    h = F.gelu(F.linear(_ln(x, w, p + 'norm2'), w[p + 'mlp.0.weight']))
    return x + F.linear(h, w[p + 'mlp.2.weight'])


def _mha(q, k, v, w, p, kv_mask=None):
    """MultiHeadAttention, transformer.py:55-72: biased q/k/v projections,
    linear attention (kv_mask: the decoder's memory_mask, :244-249; its q_mask
    = tgt_mask is always None, :361-381), bias-free merge."""
    Q = _heads(F.linear(q, w[p + 'q_proj.weight'], w[p + 'q_proj.bias']))
    K = _heads(F.linear(k, w[p + 'k_proj.weight'], w[p + 'k_proj.bias']))
    V = _heads(F.linear(v, w[p + 'v_proj.weight'], w[p + 'v_proj.bias']))
    out = linear_attention(Q, K, V, kv_mask=kv_mask).reshape(q.shape)
    return F.linear(out, w[p + 'merge.weight'])


def decoder_layer(tgt, memory, tgt_pos, m_pos, w, p, memory_mask=None):
    """transformer.py:224-255 in eval mode (dropouts are identity):
    self-attention on the query token (q = k = LN1(tgt)+tgt_pos, v = LN1(tgt));
    cross-attention with k = memory+m_pos and v = memory - NO position on v
    and NO norm on memory (:240-246); ReLU MLP (:208-212)."""
    t2 = _ln(tgt, w, p + 'norm1')
    qk = t2 + tgt_pos
    tgt = tgt + _mha(qk, qk, t2, w, p + 'self_attn.')
    t2 = _ln(tgt, w, p + 'norm2')
    tgt = tgt + _mha(t2 + tgt_pos, memory + m_pos, memory, w,
                     p + 'multihead_attn.', kv_mask=memory_mask)
    t2 = _ln(tgt, w, p + 'norm3')
    t2 = F.linear(F.relu(F.linear(t2, w[p + 'mlp.0.weight'])),
                  w[p + 'mlp.2.weight'])
    return tgt + t2


def encoder_stack(x1, x2, p1, p2, w, n_layers=N_ENC, attention=linear_attention,
                  mask1=None, mask2=None):
    """transformer.py:349-358: even layers self, odd layers cross; in a cross
    layer both images read the other's PRE-update features (:354-356).
    mask1 / mask2 [N,L1] / [N,L2] (flattened, :340-343) or None."""
    for i in range(n_layers):
        p = f'transformer.encoder.{i}.'
        if i % 2 == 0:
            x1 = encoder_layer(x1, x1, p1, p1, w, p, attention, mask1, mask1)
            x2 = encoder_layer(x2, x2, p2, p2, w, p, attention, mask2, mask2)
        else:
            y1 = encoder_layer(x1, x2, p1, p2, w, p, attention, mask1, mask2)
            y2 = encoder_layer(x2, x1, p2, p1, w, p, attention, mask2, mask1)
            x1, x2 = y1, y2
    return x1, x2


def tokens(t_nchw):
    """[N,C,h,w] -> [N,h*w,C] (transformer.py:338-345)."""
    return t_nchw.flatten(2).permute(0, 2, 1)


def feature_correlation(feat1, feat2, pos1, pos2, w, n_enc_layers=N_ENC,
                        attention=linear_attention, mask1=None, mask2=None):
    """model.py:132-143 -> QueryTransformer.forward, transformer.py:313-383.
    Returns hs1, hs2 [N,1,C] and memory1 [N,L1,C], memory2 [N,L2,C].
    mask1 / mask2 [N,hf,wf] (any numeric type; flattened :340-343): x_mask /
    source_mask of every encoder layer (:349-358) and memory_mask of the
    decoder's cross-attention (:361-381)."""
    x1, x2 = tokens(feat1), tokens(feat2)
    p1, p2 = tokens(pos1), tokens(pos2)
    n = x1.shape[0]
    if mask1 is not None:
        mask1 = mask1.flatten(1)
    if mask2 is not None:
        mask2 = mask2.flatten(1)
    x1, x2 = encoder_stack(x1, x2, p1, p2, w, n_enc_layers, attention, mask1, mask2)
    hs = []
    for mem, mpos, qe, mm in ((x1, p1, w['query_embed1.weight'], mask1),
                              (x2, p2, w['query_embed2.weight'], mask2)):
        qpos = qe.unsqueeze(0).repeat(n, 1, 1)
        tgt = torch.zeros_like(qpos)
        for i in range(N_DEC):
            tgt = decoder_layer(tgt, mem, qpos, mpos, w,
                                f'transformer.decoder.layers.{i}.', mm)
        hs.append(tgt)
    return hs[0], hs[1], x1, x2


# --------------------------------------------------------------------------
# heads (reference src/model.py)
# --------------------------------------------------------------------------
def heatmap_logits(hs, memory, hf, wf, w):
    """model.py:147-164: att = memory.hs^T; conv3x3 -> GroupNorm(32) -> ReLU
    -> conv1x1 on (memory*att) laid out NCHW; returns [N,L] logits
    (softmax_temperature = 1, model.py:95)."""
    n, L, c = memory.shape
    att = torch.einsum('blc,bnc->bln', memory, hs)
    hm = (memory * att).permute(0, 2, 1).reshape(n, c, hf, wf)
    y = F.conv2d(hm, w['heatmap_conv.0.weight'], w['heatmap_conv.0.bias'],
                 padding=1)
    y = F.group_norm(y, GN_GROUPS, w['heatmap_conv.1.weight'],
                     w['heatmap_conv.1.bias'], 1e-5)
    y = F.conv2d(F.relu(y), w['heatmap_conv.3.weight'],
                 w['heatmap_conv.3.bias'])
    return y.reshape(n, L)


def soft_argmax(logits, hf, wf, img_h):
    """model.py:173-184 + generate_mesh_grid :103-107.  kornia's
    ``create_meshgrid(h, w, normalized=False)`` is [1,h,w,2] with (x,y) in the
    last dim, x = 0..w-1, y = 0..h-1 (kornia is un-vendored and unpinned in the
    reference's requirements.txt; semantics pinned by the golden cxy vectors).
    ``stride = img_h // hf`` scales BOTH axes (:176-181)."""
    stride = img_h // hf
    prob = torch.softmax(logits, dim=1).unsqueeze(-1)          # [N,L,1]
    ys, xs = torch.meshgrid(torch.arange(hf), torch.arange(wf), indexing='ij')
    grid = torch.stack([xs, ys], dim=-1).reshape(1, hf * wf, 2).to(logits.dtype)
    coord = (grid + 0.5) * stride
    return (prob * coord).sum(1)                               # [N,2] (x,y)


MASK_FILL = -1e9   # model.py:22 (INF = 1e9), :166-171


def mask_logits(logits, mask):
    """model.py:166-171: ``heatmap_flatten.masked_fill_(~mask.flatten(1).bool(), -INF)``."""
    if mask is None:
        return logits
    return logits.masked_fill(~mask.flatten(1).bool(), MASK_FILL)


def center_estimation(hs1, hs2, memory1, memory2, hf1, wf1, hf2, wf2,
                      img_h1, img_h2, w, mask1=None, mask2=None):
    """model.py:145-186."""
    c1 = soft_argmax(mask_logits(heatmap_logits(hs1, memory1, hf1, wf1, w), mask1), hf1, wf1, img_h1)
    c2 = soft_argmax(mask_logits(heatmap_logits(hs2, memory2, hf2, wf2, w), mask2), hf2, wf2, img_h2)
    return c1, c2


def size_regression(hs, w):
    """model.py:188-191 / tlbr_reg :59-63 for one image side: [N,1,C]->[N,4]
    (top,left,bottom,right fractions)."""
    h = F.relu(F.linear(hs, w['tlbr_reg.0.weight']))
    return torch.sigmoid(F.linear(h, w['tlbr_reg.2.weight'],
                                  w['tlbr_reg.2.bias'])).squeeze(1)


def box_tlbr_to_xyxy(cxy, tlbr, max_h, max_w):
    """src/models/utils.py:16-28: centre -+ fractional extents, clamped."""
    t, l, b, r = tlbr.unbind(-1)
    x, y = cxy.unbind(-1)
    x1 = (x - l * max_w).clamp(min=0.0, max=max_w)
    y1 = (y - t * max_h).clamp(min=0.0, max=max_h)
    x2 = (x + r * max_w).clamp(min=0.0, max=max_w)
    y2 = (y + b * max_h).clamp(min=0.0, max=max_h)
    return torch.stack([x1, y1, x2, y2], dim=-1)


def bbox_iou_aligned(a, b, eps=1e-6):
    """Aligned IoU, src/losses/utils.py:69-104 (``union = max(union, eps)``)."""
    lt = torch.max(a[:, :2], b[:, :2])
    rb = torch.min(a[:, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    union = torch.clamp(area_a + area_b - inter, min=eps)
    return inter / union


def bbox_iou_matrix(a, b, eps=1e-6):
    """Un-aligned IoU [m,n], src/losses/utils.py:86-98."""
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    union = torch.clamp(area_a[:, None] + area_b - inter, min=eps)
    return inter / union


# --------------------------------------------------------------------------
# whole hot path: feature maps -> boxes (model.py:229-252 minus extraction)
# --------------------------------------------------------------------------
def cast_weights(w, dtype):
    return {k: v.to(dtype) for k, v in w.items()}


@torch.no_grad()
def hot_path(feat1, feat2, w, img_hw1, img_hw2, pos1=None, pos2=None,
             return_stages=False, attention=None, mask1=None, mask2=None):
    """feat [N,256,hf,wf] -> (box1, box2) [N,4] xyxy pixels, following
    OETR.forward_dummy after feature_extraction (model.py:239-252); mask1 /
    mask2 [N,hf,wf] = forward_dummy's optional masks (:229).  With masks the
    `logits` stage holds the filled values (-1e9 at masked tokens)."""
    dtype = feat1.dtype
    hf1, wf1 = feat1.shape[2:]
    hf2, wf2 = feat2.shape[2:]
    if pos1 is None:
        pos1 = position_table(hf1, wf1, dtype=dtype)
    if pos2 is None:
        pos2 = position_table(hf2, wf2, dtype=dtype)
    hs1, hs2, m1, m2 = feature_correlation(feat1, feat2, pos1, pos2, w,
                                           attention=attention or linear_attention,
                                           mask1=mask1, mask2=mask2)
    lg1 = mask_logits(heatmap_logits(hs1, m1, hf1, wf1, w), mask1)
    lg2 = mask_logits(heatmap_logits(hs2, m2, hf2, wf2, w), mask2)
    c1 = soft_argmax(lg1, hf1, wf1, img_hw1[0])
    c2 = soft_argmax(lg2, hf2, wf2, img_hw2[0])
    t1, t2 = size_regression(hs1, w), size_regression(hs2, w)
    b1 = box_tlbr_to_xyxy(c1, t1, img_hw1[0], img_hw1[1])
    b2 = box_tlbr_to_xyxy(c2, t2, img_hw2[0], img_hw2[1])
    if return_stages:
        return dict(hs1=hs1, hs2=hs2, memory1=m1, memory2=m2, logits1=lg1,
                    logits2=lg2, cxy1=c1, cxy2=c2, tlbr1=t1, tlbr2=t2,
                    box1=b1, box2=b2)
    return b1, b2
