import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, numpy as np
import safe_grid_agents_amd as S
from test_gpu_deepq import _args
torch.manual_seed(3)
env = S.BatchedGridworldEnv("SideEffectsSokoban-v0", 4096, seed=9, layout="compact")
env.bind_torch_stream()
env.step_random(23, auto_reset=True)
agent = S.BatchedDeepQAgent(env, _args())
scores = agent.scores().cpu().numpy()
cpu_net = agent.build_Q(36, 2, 100)
cpu_net.load_state_dict({k: v.cpu() for k, v in agent.Q.state_dict().items()})
obs = torch.as_tensor(env.boards_host().reshape(4096, -1).astype(np.float32))
with torch.no_grad():
    want = cpu_net(obs).numpy()
print("allclose", np.allclose(scores, want, rtol=1e-4, atol=1e-4))
r = agent.act().cpu().numpy()
print("act hist", np.bincount(r, minlength=4), "want hist", np.bincount(want.argmax(1), minlength=4))
print("fw w1t sum", float(agent._fw["w1t"].abs().sum()), "W1 sum", float(agent.Q[0][0].weight.abs().sum()), "stale", agent._fw_stale)
