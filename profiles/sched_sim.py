"""List-scheduling simulation behind the longest-work-first tile order: downloads the tile x keyframe
masks of the C3 scene (pcp_tile_masks) and compares the makespan of cloud order with 4-tile workgroups against
longest-first order with 1-tile workgroups.   python profiles/sched_sim.py   (on an MI355X)"""
import os
import sys, heapq, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessor_amd import capi, synth
ctx = capi.Context(0)
ctx.set_camera(capi.camera_from_dict(synth.camera_dict("cfg")))
x, y, z, _ = synth.make_cloud(10_000_000)
ctx.upload_cloud(x, y, z)
poses, _ = synth.make_trajectory(256)
ctx.set_frames(poses)
ctx.depth_pass()
m = ctx.tile_masks()
work = np.array([bin(int(v)).count("1") for v in m.reshape(-1)], dtype=np.int64).reshape(m.shape).sum(1) if False else np.unpackbits(m.view(np.uint8), axis=1).sum(1).astype(np.int64)
print("tiles", len(work), "mean", work.mean(), "max", work.max(), "p99", np.percentile(work, 99), "zero", (work == 0).mean())
blocks = work[: len(work) // 4 * 4].reshape(-1, 4).max(1) + 0.5  # a block lasts as long as its slowest wave (+ fixed cost)
def makespan(order, slots=2048):
    h = [0.0] * slots
    heapq.heapify(h)
    for b in order:
        t = heapq.heappop(h)
        heapq.heappush(h, t + blocks[b])
    return max(h)
n = len(blocks)
ideal = blocks.sum() / 2048
print("ideal", ideal, "in-order", makespan(range(n)) / ideal, "lpt", makespan(np.argsort(-blocks)) / ideal, "random", makespan(np.random.default_rng(0).permutation(n)) / ideal)
tw = work + 0.5
def makespan2(costs, order, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for b in order:
        t = heapq.heappop(h)
        heapq.heappush(h, t + costs[b])
    return max(h)
ideal_w = tw.sum() / 8192
print("wave-blocks: ideal", ideal_w, "vs 4-wave ideal", ideal * 2048 / 8192 * 4 / 4, "in-order", makespan2(tw, range(len(tw)), 8192) / ideal_w, "lpt", makespan2(tw, np.argsort(-tw), 8192) / ideal_w)
print("4-wave in-order / wave-lpt:", makespan(range(n)) * 1.0 / (makespan2(tw, np.argsort(-tw), 8192)))
