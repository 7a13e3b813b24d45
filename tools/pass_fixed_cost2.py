import sys, os, torch
sys.path.insert(0, "/root/repo")
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
B,H,W=768,16,16
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
seg = Segmenter()
ws = seg._workspace(B, H, W, "global")
seg.ops.gabor_features(imgs, ws["feats"])
seg.ops.kmeans_init(ws["feats"], B, H, W, 8, 1, ws["cent"])
f = lambda: seg.ops.assign_accumulate(ws["feats"], ws["cent"], B, H, W, 8, 1, ws["labels"], ws["partials"])
for _ in range(3): f()
torch.cuda.synchronize()
# back-to-back launches: time 50 launches as a whole
s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): f()
e.record(); torch.cuda.synchronize()
print(os.environ.get("GCS_LIB_PATH","default").split("/")[-1], "per launch back-to-back %.2f us"%(s.elapsed_time(e)*1e3/50))
