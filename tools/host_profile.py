import os, sys, cProfile, pstats, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gripnet_amd.pipeline import PoseModel, PoseStages
from gripnet_amd.synth import make_pose
dev = torch.device("cuda:0")
data = make_pose("pose0-syn").to(dev)
torch.manual_seed(1111)
model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
with torch.no_grad():
    st = PoseStages(model, data, graphs=False)
    for _ in range(10): st.step()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): st.step()
    pr.disable(); torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(22)
