"""Shared helpers for the tests: synthetic policies and the board-transpose twist."""
import numpy as np


def puzzle_transpose_twist(n):
    """Twist set {identity, transpose} for an n x n sliding puzzle in the data contract of
    docs/twists.md: obs id (pos, val) -> (T(pos), T(val)); actions left<->up, right<->down.
    Puzzle itself returns no twists (env.rs:59), so BASELINE config 3 supplies these through
    the Policy constructor."""
    n2 = n * n
    T = [(i % n) * n + (i // n) for i in range(n2)]
    ident = list(range(n2 * n2))
    tr = [T[o // n2] * n2 + T[o % n2] for o in range(n2 * n2)]
    return [ident, tr], [[0, 1, 2, 3], [1, 0, 3, 2]]


def make_policy_arrays(n2, seed=0, emb=512, hidden=256, n_actions=4, scale=1.0):
    """Synthetic BasicPolicy weights in the reference's export layout
    (src/twisterl/nn/utils.py:17-79): torch.nn.Linear default init U(-1/sqrt(fan_in), ..)."""
    rng = np.random.default_rng(seed)
    obs_size = n2 * n2

    def lin(i, o):
        b = scale / np.sqrt(i)
        w = rng.uniform(-b, b, size=(o, i)).astype(np.float32)      # torch layout [out][in]
        bias = rng.uniform(-b, b, size=o).astype(np.float32)
        return w, bias

    we, be = lin(obs_size, emb)
    w1, b1 = lin(emb, hidden)
    wa, ba = lin(hidden, n_actions)
    wv, bv = lin(hidden, 1)
    return (np.ascontiguousarray(we.T), be,
            [(np.ascontiguousarray(w1.T).reshape(-1), b1, True)],
            [(np.ascontiguousarray(wa.T).reshape(-1), ba, False)],
            [(np.ascontiguousarray(wv.T).reshape(-1), bv, False)])


def make_deep_policy_arrays(n2, seed=0, emb=64, common=(128, 64), policy_layers=(), value_layers=(), n_actions=4, scale=1.0):
    """Weights of a BasicPolicy with any number of common / policy / value layers (src/twisterl/nn/policy.py:60-113 builds
    them with make_sequential: every hidden Linear is followed by ReLU, the final action / value Linear is not) in the
    reference's export layout (src/twisterl/nn/utils.py:17-42)."""
    rng = np.random.default_rng(seed)

    def lin(i, o, relu):
        b = scale / np.sqrt(i)
        w = rng.uniform(-b, b, size=(o, i)).astype(np.float32)
        bias = rng.uniform(-b, b, size=o).astype(np.float32)
        return (np.ascontiguousarray(w.T).reshape(-1), bias, relu)

    b0 = scale / np.sqrt(n2 * n2)
    we = rng.uniform(-b0, b0, size=(emb, n2 * n2)).astype(np.float32)
    be = rng.uniform(-b0, b0, size=emb).astype(np.float32)
    cs, w = [], emb
    for h in common:
        cs.append(lin(w, h, True)); w = h
    acts, wa = [], w
    for h in policy_layers:
        acts.append(lin(wa, h, True)); wa = h
    acts.append(lin(wa, n_actions, False))
    vals, wv = [], w
    for h in value_layers:
        vals.append(lin(wv, h, True)); wv = h
    vals.append(lin(wv, 1, False))
    return (np.ascontiguousarray(we.T), be, cs, acts, vals)


def amd_policy(arrs, obs_perms=(), act_perms=()):
    """twisterl_amd.nn.Policy from make_policy_arrays() output, built through the same
    constructor calls BasicPolicy.to_rust() makes (src/twisterl/nn/policy.py:191-199)."""
    from twisterl_amd import twisterl
    emb, eb, common, action, value = arrs
    seq = lambda ls: twisterl.nn.Sequential([twisterl.nn.Linear(w.tolist(), b.tolist(), r) for (w, b, r) in ls])
    return twisterl.nn.Policy(twisterl.nn.EmbeddingBag(emb.tolist(), eb.tolist(), True, [emb.shape[0]], 0),
                              seq(common), seq(action), seq(value), [list(p) for p in obs_perms],
                              [list(p) for p in act_perms])


def oracle_policy(oracle, arrs, obs_perms=(), act_perms=()):
    return oracle.Policy(*arrs, obs_perms, act_perms)


def f32_bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def trained_puzzle8_arrays():
    """The reference's trained Puzzle-8 checkpoint (examples/ppo_puzzle8_v1.pt -> tests/golden/ppo_puzzle8_v1_weights.npz,
    written by scripts/make_trained_fixture.py) exported the way BasicPolicy.to_rust() does (src/twisterl/nn/utils.py:17-59,
    nn/policy.py:191-199): Linear weights = torch_weight.T.flatten(), EmbeddingBag vectors = torch_weight.T."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ppo_puzzle8_v1_weights.npz"))
    g = lambda k: np.ascontiguousarray(z[k.replace(".", "__")], dtype=np.float32)
    lin = lambda name, relu: (np.ascontiguousarray(g(name + ".weight").T).reshape(-1), g(name + ".bias"), relu)
    return (np.ascontiguousarray(g("embeddings.weight").T), g("embeddings.bias"),
            [lin("common.0", True)], [lin("action.0", False)], [lin("value.0", False)])


def build_stub_rccl():
    """tests/stub_rccl.hip -> tests/_build/libstub_rccl.so (hipcc, gfx950): the host-staged stand-in for RCCL that lets 2 and 3
    ranks exchange through tw_comm_* / tw_gather_* on one GPU.  Test infrastructure; selected with TW_RCCL_LIBRARY."""
    import os
    import subprocess
    from twisterl_amd.build import hipcc
    here = os.path.dirname(os.path.abspath(__file__))
    src, out = os.path.join(here, "stub_rccl.hip"), os.path.join(here, "_build", "libstub_rccl.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        cmd = [hipcc(), "-O2", "--offload-arch=gfx950", "-Wall", "-fPIC", "-shared", src, "-o", out + ".tmp", "-lpthread"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on tests/stub_rccl.hip:\n" + r.stdout)
        os.replace(out + ".tmp", out)
    return out
