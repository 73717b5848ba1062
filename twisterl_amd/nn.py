"""`twisterl.nn` surface: Linear, EmbeddingBag, Sequential, Policy.

Mirrors reference rust/src/python_interface/{layers.rs:19-49, modules.rs:19-33, policy.rs:20-46}.
The constructors take exactly what `BasicPolicy.to_rust()` passes (src/twisterl/nn/policy.py:
191-199, src/twisterl/nn/utils.py:17-79), so the reference's Python policy exports unchanged.
Weights are held on the host until first use, then uploaded once to the GPU
(tw_policy_create); all arithmetic runs in HIP kernels -- without a GPU every compute method
raises RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib


class Linear:
    """Linear(weights_vector, bias_vector, apply_relu) (layers.rs:26-33 / nn/layers.rs:24-29):
    weights_vector = torch_weight.T.flatten(), i.e. [in][out] row-major."""

    def __init__(self, weights_vector, bias_vector, apply_relu):
        self.bias = np.ascontiguousarray(bias_vector, dtype=np.float32).reshape(-1)
        self.weights = np.ascontiguousarray(weights_vector, dtype=np.float32).reshape(-1)
        if self.bias.size == 0 or self.weights.size % self.bias.size != 0:
            raise ValueError("Linear: len(weights_vector) must be a multiple of len(bias_vector)")
        self.out_features = int(self.bias.size)
        self.in_features = int(self.weights.size // self.bias.size)
        self.apply_relu = bool(apply_relu)


class EmbeddingBag:
    """EmbeddingBag(vec_vectors, bias_vector, apply_relu, obs_shape, conv_dim) (layers.rs:42-48)."""

    def __init__(self, vec_vectors, bias_vector, apply_relu, obs_shape, conv_dim):
        self.vectors = np.ascontiguousarray(vec_vectors, dtype=np.float32)
        if self.vectors.ndim != 2:
            raise ValueError("EmbeddingBag: vec_vectors must be a list of equal-length vectors")
        self.bias = np.ascontiguousarray(bias_vector, dtype=np.float32).reshape(-1)
        self.apply_relu = bool(apply_relu)
        self.obs_shape = [int(s) for s in obs_shape]
        self.conv_dim = int(conv_dim)
        if len(self.obs_shape) not in (1, 2) or self.conv_dim not in (0, 1):
            raise ValueError("EmbeddingBag: obs_shape must have one or two entries and conv_dim must be 0 or 1")

    def dense_table(self) -> np.ndarray:
        """The [obs_size][emb] table the kernels gather from.  1-D mode: the vectors themselves.  Conv1d mode
        (layers.rs:63-77): id i = (row, col) of obs_shape (swapped for conv_dim 1) adds vectors[row] into the slice
        [col*v, (col+1)*v) of the output -- the same as adding a row that is vectors[row] in that slice and +0.0
        elsewhere, and `x + 0.0 == x` bit for bit (the one exception, x = -0.0, needs an exact -0.0 bias)."""
        if len(self.obs_shape) == 1:
            return self.vectors
        rows, cols = self.obs_shape
        n_vec, v = self.vectors.shape
        n_slices = self.obs_shape[1 - self.conv_dim]
        if n_vec < self.obs_shape[self.conv_dim]:
            raise ValueError("EmbeddingBag: conv1d mode needs one vector per index of obs_shape[conv_dim]")
        if self.bias.size != n_slices * v:
            raise ValueError("EmbeddingBag: conv1d bias length must be obs_shape[1 - conv_dim] * vector length")
        table = np.zeros((rows * cols, n_slices * v), dtype=np.float32)
        for i in range(rows * cols):
            r, c = divmod(i, cols)
            if self.conv_dim == 1:
                r, c = c, r
            table[i, c * v:(c + 1) * v] = self.vectors[r]
        return table


class Sequential:
    """Sequential(layers: list[Linear]) (modules.rs:26-32)."""

    def __init__(self, layers):
        self.layers = list(layers)
        for l in self.layers:
            if not isinstance(l, Linear):
                raise TypeError("Sequential: layers must be twisterl_amd.nn.Linear")


def _linear_descs(seq: Sequential, keep: list):
    arr = (_lib.LinearDesc * max(1, len(seq.layers)))()
    for i, l in enumerate(seq.layers):
        arr[i].in_features = l.in_features
        arr[i].out_features = l.out_features
        arr[i].weights = l.weights.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].bias = l.bias.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].apply_relu = int(l.apply_relu)
        keep += [l.weights, l.bias]
    keep.append(arr)
    return arr


class Policy:
    """Policy(embeddings, common, action_net, value_net, obs_perms, act_perms) (policy.rs:26-31)."""

    def __init__(self, embeddings: EmbeddingBag, common: Sequential, action_net: Sequential,
                 value_net: Sequential, obs_perms, act_perms):
        if not isinstance(embeddings, EmbeddingBag):
            raise TypeError("embeddings must be twisterl_amd.nn.EmbeddingBag")
        for s in (common, action_net, value_net):
            if not isinstance(s, Sequential):
                raise TypeError("common/action_net/value_net must be twisterl_amd.nn.Sequential")
        self.embeddings, self.common, self.action_net, self.value_net = embeddings, common, action_net, value_net
        self.obs_perms = [[int(v) for v in p] for p in obs_perms]
        self.act_perms = [[int(v) for v in p] for p in act_perms]
        if len(self.obs_perms) != len(self.act_perms):
            raise ValueError("obs_perms and act_perms must have the same length.")
        self._h = None
        self._n_actions = action_net.layers[-1].out_features if action_net.layers else 0
        self._rng = np.random.default_rng(int.from_bytes(os.urandom(8), "little"))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().tw_policy_destroy(h)
            except Exception:
                pass

    @property
    def num_actions(self) -> int:
        return self._n_actions

    @property
    def num_perms(self) -> int:
        return len(self.obs_perms)

    def _handle(self):
        """Upload on first use (tw_policy_create); raises RuntimeError without a GPU or for
        architectures the HIP path does not implement."""
        if self._h is None:
            e = self.embeddings
            table = np.ascontiguousarray(e.dense_table(), dtype=np.float32)    # conv1d mode (Conv1dPolicy): expanded here
            keep = [table]
            d = _lib.PolicyDesc()
            d.obs_size, d.emb_size = table.shape
            if e.bias.size != table.shape[1]:
                raise ValueError("EmbeddingBag: bias length must equal the embedding size")
            d.emb_vectors = table.ctypes.data_as(C.POINTER(C.c_float))
            d.emb_bias = e.bias.ctypes.data_as(C.POINTER(C.c_float))
            d.emb_apply_relu = int(e.apply_relu)
            d.n_common, d.common = len(self.common.layers), _linear_descs(self.common, keep)
            d.n_action, d.action = len(self.action_net.layers), _linear_descs(self.action_net, keep)
            d.n_value, d.value = len(self.value_net.layers), _linear_descs(self.value_net, keep)
            d.n_perms, d.n_actions = len(self.obs_perms), self._n_actions
            if self.obs_perms:
                op = np.ascontiguousarray(self.obs_perms, dtype=np.int32)
                ap = np.ascontiguousarray(self.act_perms, dtype=np.int32)
                if op.shape != (len(self.obs_perms), d.obs_size) or ap.shape != (len(self.obs_perms), d.n_actions):
                    raise ValueError("obs_perms must be [n_perms][obs_size] and act_perms [n_perms][n_actions]")
                keep += [op, ap]
                d.obs_perms = op.ctypes.data_as(C.POINTER(C.c_int32))
                d.act_perms = ap.ctypes.data_as(C.POINTER(C.c_int32))
            h = _lib.lib().tw_policy_create(C.byref(d))
            if not h:
                raise RuntimeError(_lib.last_error())
            self._h = h
        return self._h

    # ---- batched evaluation (tw_policy_evaluate) ----------------------------------------------
    def evaluate_batch(self, mode: int, obs, masks, perms=None):
        obs = np.ascontiguousarray(obs, dtype=np.int32)
        masks = np.ascontiguousarray(masks, dtype=np.uint8)
        if obs.ndim != 2 or masks.ndim != 2 or obs.shape[0] != masks.shape[0]:
            raise ValueError("obs must be [n][n_obs] and masks [n][n_actions]")
        n, n_obs = obs.shape
        if masks.shape[1] != self._n_actions:
            raise ValueError("masks must have n_actions columns")
        out_a = np.empty((n, self._n_actions), np.float32)
        out_v = np.empty((n,), np.float32)
        pp = None
        if perms is not None:
            perms = np.ascontiguousarray(perms, dtype=np.int32)
            pp = perms.ctypes.data_as(C.POINTER(C.c_int32))
        _lib.check(_lib.lib().tw_policy_evaluate(
            self._handle(), mode, _lib.TW_PREC_F32_EXACT, obs.ctypes.data_as(C.POINTER(C.c_int32)), n, n_obs,
            masks.ctypes.data_as(C.POINTER(C.c_uint8)), pp, out_a.ctypes.data_as(C.POINTER(C.c_float)),
            out_v.ctypes.data_as(C.POINTER(C.c_float))))
        return out_a, out_v

    def _single(self, mode, obs, masks, perm):
        perms = None
        if mode != _lib.TW_EVAL_FULL_PREDICT and self.obs_perms:
            if perm is None:   # get_perm_id: uniform over the twists (policy.rs:67-77)
                perm = int(self._rng.integers(len(self.obs_perms)))
            perms = [perm]
        a, v = self.evaluate_batch(mode, [list(obs)], [[1 if m else 0 for m in masks]], perms)
        return [float(x) for x in a[0]], float(v[0])

    def predict(self, obs, masks, perm=None):
        """(masked softmax probs, value) (policy.rs:33-35 -> nn/policy.rs:34-49)"""
        return self._single(_lib.TW_EVAL_PREDICT, obs, masks, perm)

    def forward(self, obs, masks, perm=None):
        """(masked logits, value) (policy.rs:38-40 -> nn/policy.rs:51-65)"""
        return self._single(_lib.TW_EVAL_FORWARD, obs, masks, perm)

    def full_predict(self, obs, masks):
        """average over all twists (policy.rs:42-44 -> nn/policy.rs:102-126)"""
        return self._single(_lib.TW_EVAL_FULL_PREDICT, obs, masks, None)


def _sync_from_torch(policy: "Policy", state) -> None:
    """See Policy.update_from_torch."""
    import torch
    if hasattr(state, "state_dict"):
        state = state.state_dict()
    e = policy.embeddings
    if len(e.obs_shape) == 2:
        # Conv1dPolicy (src/twisterl/nn/policy.py:207-266): conv_layer.weight [v][n_vec][1], no bias.  The dense
        # [emb][obs_size] Linear weight the kernels' table is made from is assembled on the device (EmbeddingBag.dense_table).
        if "conv_layer.weight" not in state:
            raise KeyError("update_from_torch: the state has no conv_layer.weight (Conv1dPolicy layout)")
        w = state["conv_layer.weight"].detach().to(device="cuda", dtype=torch.float32)
        n_vec, v = e.vectors.shape
        if tuple(w.shape) not in ((v, n_vec, 1), (v, n_vec)):
            raise ValueError(f"update_from_torch: conv_layer.weight has shape {tuple(w.shape)}, the policy was built for {(v, n_vec, 1)}")
        w = w.reshape(v, n_vec)
        rows, cols = e.obs_shape
        i = torch.arange(rows * cols, device="cuda")
        r, c = (i // cols, i % cols) if e.conv_dim == 0 else (i % cols, i // cols)
        n_slices = e.obs_shape[1 - e.conv_dim]
        dense = torch.zeros((n_slices, v, rows * cols), device="cuda", dtype=torch.float32)
        dense[c, :, i] = w.t()[r]
        state = dict(state)
        state["embeddings.weight"] = dense.reshape(n_slices * v, rows * cols)
        state["embeddings.bias"] = torch.zeros(n_slices * v, device="cuda", dtype=torch.float32)
    if not _is_basic_shape(policy):
        return _sync_generic_from_torch(policy, state)
    keys = ["embeddings.weight", "embeddings.bias", "common.0.weight", "common.0.bias", "action.0.weight", "action.0.bias",
            "value.0.weight", "value.0.bias"]
    missing = [k for k in keys if k not in state]
    if missing:
        raise KeyError(f"update_from_torch: the state has no {missing} (BasicPolicy layout: embeddings / common.0 / action.0 / value.0)")
    h = policy._handle()
    emb, hid = int(policy.embeddings.bias.size), policy.common.layers[0].out_features
    obs_size = int(np.prod(policy.embeddings.obs_shape)) if len(policy.embeddings.obs_shape) == 2 else int(policy.embeddings.vectors.shape[0])
    want = {"embeddings.weight": (emb, obs_size), "embeddings.bias": (emb,), "common.0.weight": (hid, emb),
            "common.0.bias": (hid,), "action.0.weight": (int(policy.num_actions), hid), "action.0.bias": (int(policy.num_actions),),
            "value.0.weight": (1, hid), "value.0.bias": (1,)}
    ts = []
    for k in keys:
        t = state[k].detach()
        if tuple(t.shape) != want[k]:
            raise ValueError(f"update_from_torch: {k} has shape {tuple(t.shape)}, the policy was built for {want[k]}")
        ts.append(t.to(device="cuda", dtype=torch.float32).contiguous())
    torch.cuda.current_stream().synchronize()        # the parameters are final before the library's stream reads them
    _lib.check(_lib.lib().tw_policy_update_device(h, *[C.c_void_p(t.data_ptr()) for t in ts]))
    torch.cuda.synchronize()                          # the temporaries in `ts` may be freed after this returns


def _is_basic_shape(policy: "Policy") -> bool:
    """the one-common-layer shape the MFMA engines implement (tw_api.hip is_mfma_shape); everything else is a generic stack"""
    c, a, v = policy.common.layers, policy.action_net.layers, policy.value_net.layers
    if len(c) != 1 or len(a) != 1 or len(v) != 1:
        return False
    emb, hid = int(policy.embeddings.bias.size), c[0].out_features
    obs_size = int(np.prod(policy.embeddings.obs_shape)) if len(policy.embeddings.obs_shape) == 2 else int(policy.embeddings.vectors.shape[0])
    return (emb % 32 == 0 and hid in (32, 64, 128, 256) and obs_size <= 256 and not a[0].apply_relu and not v[0].apply_relu
            and v[0].out_features == 1)


def _sync_generic_from_torch(policy: "Policy", state) -> None:
    """update_from_torch for policies of any Sequential depth: BasicPolicy builds its stacks with make_sequential
    (src/twisterl/nn/utils.py:82-93: Linear, ReLU, Linear, ReLU, ..), so the Linear layers of a stack are its even entries."""
    import torch
    stacks = (("common", policy.common.layers), ("action", policy.action_net.layers), ("value", policy.value_net.layers))
    emb = int(policy.embeddings.bias.size)
    obs_size = int(np.prod(policy.embeddings.obs_shape)) if len(policy.embeddings.obs_shape) == 2 else int(policy.embeddings.vectors.shape[0])
    want = [("embeddings.weight", (emb, obs_size)), ("embeddings.bias", (emb,))]
    for name, layers in stacks:
        for i, lay in enumerate(layers):
            want.append((f"{name}.{2 * i}.weight", (lay.out_features, lay.in_features)))
            want.append((f"{name}.{2 * i}.bias", (lay.out_features,)))
    missing = [k for k, _ in want if k not in state]
    if missing:
        raise KeyError(f"update_from_torch: the state has no {missing} (BasicPolicy layout: embeddings, common.0/2/.., action.0/2/.., value.0/2/..)")
    ts = []
    for k, shape in want:
        t = state[k].detach()
        if tuple(t.shape) != shape:
            raise ValueError(f"update_from_torch: {k} has shape {tuple(t.shape)}, the policy was built for {shape}")
        ts.append(t.to(device="cuda", dtype=torch.float32).contiguous())
    n = (len(ts) - 2) // 2
    ws = (C.c_void_p * max(n, 1))(*[ts[2 + 2 * i].data_ptr() for i in range(n)])
    bs = (C.c_void_p * max(n, 1))(*[ts[3 + 2 * i].data_ptr() for i in range(n)])
    torch.cuda.current_stream().synchronize()
    _lib.check(_lib.lib().tw_policy_update_device_layers(policy._handle(), C.c_void_p(ts[0].data_ptr()), C.c_void_p(ts[1].data_ptr()), ws, bs, n))
    torch.cuda.synchronize()


def _update_from_torch(self, state) -> "Policy":
    """Device-to-device policy sync: refresh every weight image of this policy from a torch
    module / state_dict in the reference's BasicPolicy layout (embeddings, common.0, action.0, value.0; Linear.weight =
    [out][in]) without leaving the GPU.  Replaces rebuilding the policy from `.cpu().numpy().tolist()` exports on
    every training iteration (reference src/twisterl/nn/policy.py:191-199, rl/algorithm.py:90-93).  Shapes, ReLU
    flags and twists stay those the policy was created with."""
    _sync_from_torch(self, state)
    return self


Policy.update_from_torch = _update_from_torch
