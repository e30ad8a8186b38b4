"""Device memory in use before / after streaming many frames through the native pipeline: the stage threads' workspaces must not grow with the frame count."""
import sys, numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
import __graft_entry__ as graft
from importlib import import_module
pkg = graft.load_package()
pl = import_module(pkg.__name__ + ".pipeline")
dev = torch.device("cuda:0")
T, ndb = 1024, 300000
frames = [torch.from_numpy(pkg.synth.make_tile(T, T, frame_index=i)).to(dev) for i in range(3)]
db = torch.zeros((ndb, 64), dtype=torch.uint8, device=dev)
db[:, :61] = torch.from_numpy(pkg.synth.make_descriptor_db(ndb)).to(dev)
db_xy = torch.zeros((ndb, 2), dtype=torch.float32, device=dev)
p = pl.StreamedFramePipeline(db, db_xy)
p.run(frames, 20, filter_strength=0.3)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for rep in range(5):
    p.run(frames, 200, filter_strength=0.3)
    torch.cuda.synchronize()
    print(f"after {200 * (rep + 1)} more frames: free memory changed by {(torch.cuda.mem_get_info()[0] - free0) / 1e6:+.1f} MB", flush=True)
p.close()
