"""ORACLE (test infrastructure only) -- acoustic-model training step restatement (numpy).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this package; the product path never does.

Restates the graph built by /root/reference/models/AcousticModel.py:
  * input Linear ................ :240-250
  * stacked BasicLSTMCell + dynamic_rnn(sequence_length) ... :223-237, :266-298
  * output Linear ............... :301-309
  * label sparsification ........ :144-159 (drop id 0, fill empty rows with C-1)
  * tf.nn.ctc_loss(..., ignore_longer_outputs_than_inputs=True) ... :356-361
  * gradient accumulate / clip_by_global_norm / Adam ... :386-406
  * greedy decode stands in for the beam decoder at :312
  * calculate_wer / calculate_cer ... :530-632

The arithmetic of those ops lives in TensorFlow 1.x (un-vendored, un-pinned:
requirements.txt:2, README.md:49 says >= 1.4), which is not importable here and
has no golden vectors in the reference's tests => PARITY UNPINNED at the TF
boundary.  The TF-1.4 op semantics are restated from their published
definitions (BasicLSTMCell gate order i,j,f,o with forget_bias 1.0 added at run
time; dynamic_rnn zero-output / state copy-through past sequence_length;
ctc_loss_calculator label conventions; TF-flavoured Adam) and cross-checked in
tests/ against independent implementations (torch.nn.functional.ctc_loss and a
torch-autograd LSTM).  calculate_wer / calculate_cer ARE pinned by fixtures made
from the imported reference (tests/golden/wer_cer.json).

All functions take a `dtype` (float64 for the checker, float32 to mimic TF).
Layouts: time-major [T, B, *]; LSTM kernel K_l is [2H, 4H] (rows: x then h;
columns: i | j | f | o blocks of H).
"""
import numpy as np

FORGET_BIAS = 1.0


# ----------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------
def xavier_uniform(rng, fan_in, fan_out, dtype=np.float32):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)


def init_params(num_layers, hidden, input_dim, num_labels, seed=1234, dtype=np.float32):
    """TF defaults: Xavier/Glorot-uniform matrices, zero biases (:241-244, :302-305)."""
    rng = np.random.RandomState(seed)
    p = {"input_w": xavier_uniform(rng, input_dim, hidden, dtype),
         "input_b": np.zeros(hidden, dtype)}
    for l in range(num_layers):
        p["kernel_%d" % l] = xavier_uniform(rng, 2 * hidden, 4 * hidden, dtype)
        p["bias_%d" % l] = np.zeros(4 * hidden, dtype)
    p["output_w"] = xavier_uniform(rng, hidden, num_labels, dtype)
    p["output_b"] = np.zeros(num_labels, dtype)
    return p


def param_names(num_layers):
    names = ["input_w", "input_b"]
    for l in range(num_layers):
        names += ["kernel_%d" % l, "bias_%d" % l]
    return names + ["output_w", "output_b"]


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ----------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------
BN_EPS = 1e-3


def batch_norm(x):
    """tf.nn.moments over the batch axis + tf.nn.batch_normalization without scale/offset (:253-259)."""
    mean = x.mean(axis=1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=1, keepdims=True)
    inv = 1.0 / np.sqrt(var + BN_EPS)
    return (x - mean) * inv, inv


def batch_norm_backward(dy, xhat, inv):
    return inv * (dy - dy.mean(axis=1, keepdims=True) - xhat * (dy * xhat).mean(axis=1, keepdims=True))


def forward(p, x, lengths, num_layers, state=None, keep_cache=False,
            in_masks=None, out_masks=None, normalization=False):
    """x [T,B,D], lengths [B] -> logits [T,B,C], final state, cache.

    `in_masks[l]` / `out_masks[l]` are optional [T,B,H] inverted-dropout
    multipliers (mask/keep_prob) applied to the input and output of layer l
    (DropoutWrapper, :232-233); None == keep_prob 1.0."""
    T, B, _ = x.shape
    H = p["input_b"].shape[0]
    dt = p["input_w"].dtype
    lengths = np.asarray(lengths)
    cur = x.astype(dt) @ p["input_w"] + p["input_b"]          # [T,B,H]
    cache = {"x": x.astype(dt), "layers": []}
    if normalization:
        cur, inv = batch_norm(cur)
        cache["bn"] = (cur, inv)
    final = []
    for l in range(num_layers):
        K, bias = p["kernel_%d" % l], p["bias_%d" % l]
        if state is None:
            c = np.zeros((B, H), dt)
            h = np.zeros((B, H), dt)
        else:
            c, h = state[l][0].astype(dt).copy(), state[l][1].astype(dt).copy()
        xin = cur if in_masks is None or in_masks[l] is None else cur * in_masks[l]
        out = np.zeros((T, B, H), dt)
        gi = np.zeros((T, B, H), dt); gj = np.zeros((T, B, H), dt)
        gf = np.zeros((T, B, H), dt); go = np.zeros((T, B, H), dt)
        cs = np.zeros((T, B, H), dt); hprev = np.zeros((T, B, H), dt)
        cprev = np.zeros((T, B, H), dt)
        for t in range(T):
            live = (t < lengths)[:, None]
            g = np.concatenate([xin[t], h], axis=1) @ K + bias
            i = sigmoid(g[:, :H]); j = np.tanh(g[:, H:2 * H])
            f = sigmoid(g[:, 2 * H:3 * H] + FORGET_BIAS); o = sigmoid(g[:, 3 * H:])
            c_new = c * f + i * j
            h_new = np.tanh(c_new) * o
            hprev[t], cprev[t] = h, c
            gi[t], gj[t], gf[t], go[t], cs[t] = i, j, f, o, c_new
            out[t] = np.where(live, h_new, 0.0)
            c = np.where(live, c_new, c)
            h = np.where(live, h_new, h)
        final.append((c, h))
        y = out if out_masks is None or out_masks[l] is None else out * out_masks[l]
        if keep_cache:
            cache["layers"].append(dict(xin=xin, hprev=hprev, cprev=cprev, i=gi, j=gj,
                                        f=gf, o=go, c=cs, out=out))
        cur = y
    cache["top"] = cur
    logits = cur @ p["output_w"] + p["output_b"]
    return logits, final, cache


# ----------------------------------------------------------------------------
# bidirectional variant (BASELINE.json configs[4]; NOT in the reference, which builds a unidirectional dynamic_rnn at
# :276-278 -- restated from tf.nn.bidirectional_dynamic_rnn: a second MultiRNNCell stack runs over
# tf.reverse_sequence(inputs, sequence_length), its outputs are reversed back and concatenated with the forward ones)
# ----------------------------------------------------------------------------
def reverse_sequences(x, lengths):
    """[T,B,*]: out[t,b] = x[len_b-1-t, b] for t < len_b, 0 beyond (tf.reverse_sequence; self-adjoint)."""
    out = np.zeros_like(x)
    for b, n in enumerate(np.asarray(lengths)):
        n = int(min(max(n, 0), x.shape[0]))
        out[:n, b] = x[:n, b][::-1]
    return out


def _stack_params(p, num_layers, prefix):
    """The parameters of one direction's stack as a model whose input and output layers are identities."""
    H = p["input_b"].shape[0]
    q = {"input_w": np.eye(H, dtype=p["input_w"].dtype), "input_b": np.zeros(H, p["input_w"].dtype),
         "output_w": np.eye(H, dtype=p["input_w"].dtype), "output_b": np.zeros(H, p["input_w"].dtype)}
    for l in range(num_layers):
        q["kernel_%d" % l] = p["%skernel_%d" % (prefix, l)]
        q["bias_%d" % l] = p["%sbias_%d" % (prefix, l)]
    return q


def init_params_bidirectional(num_layers, hidden, input_dim, num_labels, seed=1234, dtype=np.float32):
    rng = np.random.RandomState(seed)
    p = {"input_w": xavier_uniform(rng, input_dim, hidden, dtype), "input_b": np.zeros(hidden, dtype)}
    for prefix in ("", "bw_"):
        for l in range(num_layers):
            p["%skernel_%d" % (prefix, l)] = xavier_uniform(rng, 2 * hidden, 4 * hidden, dtype)
            p["%sbias_%d" % (prefix, l)] = np.zeros(4 * hidden, dtype)
    p["output_w"] = xavier_uniform(rng, 2 * hidden, num_labels, dtype)
    p["output_b"] = np.zeros(num_labels, dtype)
    return p


def forward_bidirectional(p, x, lengths, num_layers, masks_fw=None, masks_bw=None):
    """-> logits [T,B,C], cache for backward_bidirectional.  masks_fw / masks_bw: optional (in_masks, out_masks) of the two
    stacks (each DropoutWrapper'd cell has its own masks; the backward-direction stack's are indexed in ITS time, i.e. on the
    reversed sequence)."""
    T, B, _ = x.shape
    dt = p["input_w"].dtype
    z0 = x.astype(dt) @ p["input_w"] + p["input_b"]
    qf, qb = _stack_params(p, num_layers, ""), _stack_params(p, num_layers, "bw_")
    mf = masks_fw if masks_fw is not None else (None, None)
    mb = masks_bw if masks_bw is not None else (None, None)
    yf, _, cf = forward(qf, z0, lengths, num_layers, keep_cache=True, in_masks=mf[0], out_masks=mf[1])
    yb_rev, _, cb = forward(qb, reverse_sequences(z0, lengths), lengths, num_layers, keep_cache=True,
                            in_masks=mb[0], out_masks=mb[1])
    yb = reverse_sequences(yb_rev, lengths)
    top = np.concatenate([yf, yb], axis=2)
    logits = top @ p["output_w"] + p["output_b"]
    return logits, dict(x=x.astype(dt), top=top, cf=cf, cb=cb, qf=qf, qb=qb, mf=mf, mb=mb)


def backward_bidirectional(p, cache, dlogits, lengths, num_layers):
    T, B, C = dlogits.shape
    H = p["input_b"].shape[0]
    dt = p["input_w"].dtype
    dl = dlogits.astype(dt).reshape(T * B, C)
    g = {"output_w": cache["top"].reshape(T * B, 2 * H).T @ dl, "output_b": dl.sum(0)}
    dtop = (dl @ p["output_w"].T).reshape(T, B, 2 * H)
    # each stack is a model with identity input / output layers: d(top) is its "dlogits"; the gradient w.r.t. its input is
    # dG_0 . K_0[:H]^T (backward() returns the gate gradients through `debug`)
    dbg_f, dbg_b = {}, {}
    mf, mb = cache.get("mf", (None, None)), cache.get("mb", (None, None))
    gf = backward(cache["qf"], cache["cf"], dtop[:, :, :H], lengths, num_layers, debug=dbg_f, in_masks=mf[0], out_masks=mf[1])
    gb = backward(cache["qb"], cache["cb"], reverse_sequences(dtop[:, :, H:], lengths), lengths, num_layers, debug=dbg_b,
                  in_masks=mb[0], out_masks=mb[1])
    for l in range(num_layers):
        g["kernel_%d" % l], g["bias_%d" % l] = gf["kernel_%d" % l], gf["bias_%d" % l]
        g["bw_kernel_%d" % l], g["bw_bias_%d" % l] = gb["kernel_%d" % l], gb["bias_%d" % l]
    d0f = dbg_f["dg_0"] @ p["kernel_0"][:H].T
    d0b = dbg_b["dg_0"] @ p["bw_kernel_0"][:H].T
    if mf[0] is not None and mf[0][0] is not None:      # the layer-0 input masks sit between Z_0 and the stacks
        d0f = d0f * mf[0][0]
    if mb[0] is not None and mb[0][0] is not None:
        d0b = d0b * mb[0][0]
    d0b = reverse_sequences(d0b, lengths)
    d0 = (d0f + d0b).reshape(T * B, H)
    g["input_w"] = cache["x"].reshape(T * B, -1).T @ d0
    g["input_b"] = d0.sum(0)
    return g


# ----------------------------------------------------------------------------
# CTC (TF ctc_loss_calculator conventions)
# ----------------------------------------------------------------------------
def sparsify_labels(dense, num_labels):
    """AcousticModel.py:155-159: keep entries != 0 in order; empty rows get [C-1]."""
    rows = []
    for r in np.asarray(dense):
        kept = [int(v) for v in r if v != 0]
        rows.append(kept if kept else [num_labels - 1])
    return rows


def ctc_targets(raw_labels, num_labels):
    """TF PopulateLPrimes: the target is every label before the first one >= C-1;
    required_time is the RAW label count (EOS included)."""
    tgt = []
    for v in raw_labels:
        if v >= num_labels - 1:
            break
        tgt.append(int(v))
    return tgt, len(raw_labels)


def _logsumexp2(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), -np.inf, r)


def ctc_loss_and_grad(logits, label_rows, lengths, blank=None):
    """logits [T,B,C]; label_rows = list of raw label lists (post-sparsify).
    Returns loss [B] and dlogits [T,B,C] (gradient of sum_b loss_b)."""
    T, B, C = logits.shape
    blank = C - 1 if blank is None else blank
    lg = logits.astype(np.float64)
    mx = lg.max(axis=2, keepdims=True)
    logp = lg - mx - np.log(np.exp(lg - mx).sum(axis=2, keepdims=True))
    y = np.exp(logp)
    loss = np.zeros(B)
    grad = np.zeros((T, B, C))
    for b in range(B):
        Tb = int(lengths[b])
        tgt, required = ctc_targets(label_rows[b], C)
        if Tb <= 0 or required > Tb:        # ignore_longer_outputs_than_inputs=True
            continue
        Tb = min(Tb, T)
        ext = np.full(2 * len(tgt) + 1, blank, dtype=np.int64)
        ext[1::2] = tgt
        S = len(ext)
        can_skip = np.zeros(S, bool)
        can_skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        lp = logp[:Tb, b, :][:, ext]                       # [Tb, S]
        alpha = np.full((Tb, S), -np.inf)
        alpha[0, 0] = lp[0, 0]
        if S > 1:
            alpha[0, 1] = lp[0, 1]
        for t in range(1, Tb):
            a = alpha[t - 1]
            acc = a.copy()
            acc[1:] = _logsumexp2(acc[1:], a[:-1])
            sk = np.full(S, -np.inf)
            sk[2:] = np.where(can_skip[2:], a[:-2], -np.inf)
            acc = _logsumexp2(acc, sk)
            alpha[t] = acc + lp[t]
        beta = np.full((Tb, S), -np.inf)                   # beta excludes y_t
        beta[Tb - 1, S - 1] = 0.0
        if S > 1:
            beta[Tb - 1, S - 2] = 0.0
        for t in range(Tb - 2, -1, -1):
            nb = beta[t + 1] + lp[t + 1]
            acc = nb.copy()
            acc[:-1] = _logsumexp2(acc[:-1], nb[1:])
            sk = np.full(S, -np.inf)
            sk[:-2] = np.where(can_skip[2:], nb[2:], -np.inf)
            beta[t] = _logsumexp2(acc, sk)
        ll = _logsumexp2(alpha[Tb - 1, S - 1], alpha[Tb - 1, S - 2] if S > 1 else -np.inf)
        loss[b] = -ll
        if np.isneginf(ll):                                # TF: "No valid path found"
            grad[:Tb, b, :] = y[:Tb, b, :]
            continue
        post = np.exp(alpha + beta - ll)                   # [Tb, S]
        occ = np.zeros((Tb, C))
        for s in range(S):
            occ[:, ext[s]] += post[:, s]
        grad[:Tb, b, :] = y[:Tb, b, :] - occ
    return loss.astype(logits.dtype), grad.astype(logits.dtype)


# ----------------------------------------------------------------------------
# backward
# ----------------------------------------------------------------------------
def backward(p, cache, dlogits, lengths, num_layers, in_masks=None, out_masks=None, debug=None):
    """BPTT for the graph in forward(); returns dict of gradients (sum over batch)."""
    T, B, C = dlogits.shape
    H = p["input_b"].shape[0]
    dt = p["input_w"].dtype
    lengths = np.asarray(lengths)
    g = {}
    top = cache["top"].reshape(T * B, H)
    dl = dlogits.astype(dt).reshape(T * B, C)
    g["output_w"] = top.T @ dl
    g["output_b"] = dl.sum(0)
    dy = (dl @ p["output_w"].T).reshape(T, B, H)
    for l in range(num_layers - 1, -1, -1):
        L = cache["layers"][l]
        K = p["kernel_%d" % l]
        if out_masks is not None and out_masks[l] is not None:
            dy = dy * out_masks[l]
        dg_all = np.zeros((T, B, 4 * H), dt)
        dxin = np.zeros((T, B, H), dt)
        dh = np.zeros((B, H), dt)
        dc = np.zeros((B, H), dt)
        for t in range(T - 1, -1, -1):
            live = (t < lengths)[:, None]
            i, j, f, o, c = L["i"][t], L["j"][t], L["f"][t], L["o"][t], L["c"][t]
            dh_tot = dh + dy[t]
            tc = np.tanh(c)
            do = dh_tot * tc
            dc_tot = dc + dh_tot * o * (1.0 - tc * tc)
            dgi = dc_tot * j * i * (1.0 - i)
            dgj = dc_tot * i * (1.0 - j * j)
            dgf = dc_tot * L["cprev"][t] * f * (1.0 - f)
            dgo = do * o * (1.0 - o)
            dg = np.concatenate([dgi, dgj, dgf, dgo], axis=1)
            dg = np.where(live, dg, 0.0)
            dg_all[t] = dg
            dxh = dg @ K.T
            dxin[t] = dxh[:, :H]
            dh = np.where(live, dxh[:, H:], dh)
            dc = np.where(live, dc_tot * f, dc)
        # dK = sum_t [x_t ; h_{t-1}]^T . dg_t, as one product over all frames (the sum is time-independent)
        xh = np.concatenate([L["xin"], L["hprev"]], axis=2).reshape(T * B, 2 * H)
        g["kernel_%d" % l] = xh.T @ dg_all.reshape(T * B, 4 * H)
        g["bias_%d" % l] = dg_all.sum(axis=(0, 1))
        if debug is not None:          # tests: the gate gradients of every frame, for localising a mismatch
            debug["dg_%d" % l] = dg_all
        if in_masks is not None and in_masks[l] is not None:
            dxin = dxin * in_masks[l]
        dy = dxin
    if "bn" in cache:
        dy = batch_norm_backward(dy, *cache["bn"])
    d0 = dy.reshape(T * B, H)
    g["input_w"] = cache["x"].reshape(T * B, -1).T @ d0
    g["input_b"] = d0.sum(0)
    return g


# ----------------------------------------------------------------------------
# optimiser (tf.clip_by_global_norm + tf.train.AdamOptimizer), :386-406
# ----------------------------------------------------------------------------
def global_norm(grads):
    return float(np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in grads.values())))


def clip_and_adam(p, grads, m, v, step, lr, clip, beta1=0.9, beta2=0.999, eps=1e-8):
    """In place. `step` is the 1-based Adam step count after this update."""
    gn = global_norm(grads)
    scale = clip / max(gn, clip)
    lr_t = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    for k in p:
        gk = grads[k] * p[k].dtype.type(scale)
        m[k] = beta1 * m[k] + (1.0 - beta1) * gk
        v[k] = beta2 * v[k] + (1.0 - beta2) * gk * gk
        p[k] -= (lr_t * m[k] / (np.sqrt(v[k]) + eps)).astype(p[k].dtype)
    return gn


def train_step(p, m, v, step, batches, num_layers, lr, clip, state=None, carry_state=True):
    """One optimiser step over k mini-batches (run_train_step, :887-939), dropout off.
    batches: list of (x[T,B,D], lengths[B], dense_labels[B,U]).  The reference carries the
    RNN state from one mini-batch into the next (rnn_keep_state_op is fetched by every
    run_step, :642); carry_state=False restarts every mini-batch from `state` (the data-
    parallel equivalence oracle).  Returns (mean logged loss, grads, state)."""
    C = p["output_b"].shape[0]
    acc = {k: np.zeros_like(a) for k, a in p.items()}
    logged = 0.0
    for x, lengths, dense in batches:
        rows = sparsify_labels(dense, C)
        logits, new_state, cache = forward(p, x, lengths, num_layers, state=state, keep_cache=True)
        if carry_state:
            state = new_state
        loss, dlogits = ctc_loss_and_grad(logits, rows, lengths)
        gr = backward(p, cache, dlogits, lengths, num_layers)
        for k in acc:
            acc[k] += gr[k]
        with np.errstate(divide="ignore", invalid="ignore"):
            logged += float(np.mean(loss / np.asarray(lengths, dtype=np.float64)))  # :361
    clip_and_adam(p, acc, m, v, step, lr, clip)
    return logged / len(batches), acc, state


# ----------------------------------------------------------------------------
# decode + metrics
# ----------------------------------------------------------------------------
def greedy_decode(logits, lengths, blank=None):
    """argmax per frame -> collapse repeats -> drop blank.  List of id lists."""
    T, B, C = logits.shape
    blank = C - 1 if blank is None else blank
    best = logits.argmax(axis=2)
    out = []
    for b in range(B):
        prev = -1
        ids = []
        for t in range(min(int(lengths[b]), T)):
            k = int(best[t, b])
            if k != prev and k != blank:
                ids.append(k)
            prev = k
        out.append(ids)
    return out


def edit_distance(a, b):
    """Levenshtein distance between two sequences (tf.edit_distance un-normalised)."""
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cur[j] = min(prev[j - 1] + (a[i - 1] != b[j - 1]), cur[j - 1] + 1, prev[j] + 1)
        prev = cur
    return prev[len(b)]


def calculate_wer(first, second):
    """AcousticModel.calculate_wer (:530-582): word-level distance; the reference
    stores the table as uint8, so the result wraps modulo 256."""
    return edit_distance(first.split(), second.split()) % 256


def calculate_cer(first, second):
    """AcousticModel.calculate_cer (:584-632): spaces stripped, uint16 table."""
    return edit_distance(first.replace(" ", ""), second.replace(" ", "")) % 65536
