"""End-to-end sketch of the reference's PPO iteration with the MI355X collector in the loop (needs a GPU):

    collect (HIP kernels) -> data_to_torch (HIP kernels, tensors stay on the GPU) -> PPO update (plain torch, as in the
    reference's trainer) -> policy sync (device-to-device) -> evaluate (HIP kernels)

This mirrors Algorithm.learn_step / PPO.train_step of the reference (src/twisterl/rl/algorithm.py:105-125, rl/ppo.py:63-110)
closely enough to show that nothing in the loop goes through host lists; it is NOT the trainer (out of scope here).
"""
import sys
import torch
sys.path.insert(0, ".")
from twisterl_amd import twisterl, trainer


class BasicPolicyTwin(torch.nn.Module):
    """torch twin of the policy the collector runs (reference src/twisterl/nn/policy.py:115-120): the same layer names."""

    def __init__(self, obs_size, emb=128, hidden=64, n_actions=4):
        super().__init__()
        self.embeddings = torch.nn.Linear(obs_size, emb)
        self.common = torch.nn.Sequential(torch.nn.Linear(emb, hidden), torch.nn.ReLU())
        self.action = torch.nn.Sequential(torch.nn.Linear(hidden, n_actions))
        self.value = torch.nn.Sequential(torch.nn.Linear(hidden, 1))

    def forward(self, x):
        h = self.common(torch.relu(self.embeddings(x)))
        return self.action(h), self.value(h).squeeze(-1)

    def to_collector(self):
        """what BasicPolicy.to_rust() does (nn/policy.py:191-199, nn/utils.py:17-79), once"""
        lin = lambda l, relu: twisterl.nn.Linear(l.weight.detach().T.flatten().tolist(), l.bias.detach().tolist(), relu)
        e = self.embeddings
        return twisterl.nn.Policy(
            twisterl.nn.EmbeddingBag(e.weight.detach().T.tolist(), e.bias.detach().tolist(), True, [e.in_features], 0),
            twisterl.nn.Sequential([lin(self.common[0], True)]), twisterl.nn.Sequential([lin(self.action[0], False)]),
            twisterl.nn.Sequential([lin(self.value[0], False)]), [], [])


def run(iterations=3, episodes=8192, precision="fp32", seed=0, log=print):
    torch.manual_seed(seed)
    side, obs_size = 3, 81
    model = BasicPolicyTwin(obs_size).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=3e-3)
    env = twisterl.env.Puzzle(side, side, 3, 4, 256)
    rs_pol = model.cpu().to_collector(); model.cuda()
    coll = twisterl.collector.PPOCollector(**{"num_episodes": episodes, "gamma": 0.995, "lambda": 0.995, "num_cores": 32},
                                           precision=precision)
    history = []
    for it in range(iterations):
        data = coll.collect(env, rs_pol, seed=seed + it)
        pt_obs, old_logp, acts, advs, rets, _ = trainer.ppo_data_to_torch(data, obs_size, normalize_advantage=True)
        for _ in range(4):                                          # PPO clip objective (rl/ppo.py:63-110)
            logits, vals = model(pt_obs)
            logp = torch.distributions.Categorical(logits=logits).log_prob(acts)
            ratio = torch.exp(logp - old_logp)
            loss = -torch.min(ratio * advs, torch.clamp(ratio, 0.8, 1.2) * advs).mean() + 0.5 * (vals - rets).pow(2).mean()
            opt.zero_grad(); loss.backward(); opt.step()
        rs_pol.update_from_torch(model)                              # device-to-device policy sync
        succ, rew = twisterl.collector.evaluate(env, rs_pol, 256, True, 1, 0, seed, 1.41, 1, 32)
        history.append((loss.item(), succ, rew, len(data)))
        log(f"iter {it}: records {len(data)} loss {loss.item():.4f} success {succ:.3f} reward {rew:.3f}")
    return history


if __name__ == "__main__":
    run()
