"""where an eager capacity-mode frame spends its time around the conv stack: the two stack brackets (level 1 | the rest; waits for the levels' geometry
included) per model, and the frame.   python tools/probe_stack_brackets.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench


def main():
    from lidarseg3d_amd import ops, scn_unet, synth
    dev = torch.device("cuda:0")
    ops.set_precision("bf16x6")
    for kind in ("sdseg3d", "mseg3d"):
        model, _ = bench.build_model(dev, kind=kind)
        f = synth.lidar_frame(120000, seed=100, **synth.NUSC)
        pts = torch.from_numpy(np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)).to(dev)
        ex = dict(points=pts, batch_size=1)
        if kind == "mseg3d":
            img, emb, cuv = synth.camera_inputs(120000, seed=100, ncam=6, c_img=48, h=160, w=240, batch=1)
            ex.update(points_cuv=torch.from_numpy(cuv).to(dev), image_features=torch.from_numpy(img).to(dev), camera_semantic_embeddings=torch.from_numpy(emb).to(dev))
        events = []
        scn_unet.UNetSCN3D.conv_stack_events = events
        frames = []
        with torch.no_grad():
            for i in range(13):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if i == 3:
                    del events[:]
                s.record()
                model(dict(ex), return_loss=False)
                e.record()
                if i >= 3:
                    frames.append((s, e))
        torch.cuda.synchronize()
        scn_unet.UNetSCN3D.conv_stack_events = None
        per = len(events) // len(frames)
        b = [[a.elapsed_time(z) for a, z in events[i::per]] for i in range(per)]
        fr = [a.elapsed_time(z) for a, z in frames]
        first = [frames[i][0].elapsed_time(events[i * per][0]) for i in range(len(frames))]
        print("%-8s frame %.3f ms | frame start -> stack start %.3f | brackets %s | sum %.3f" % (
            kind, statistics.median(fr), statistics.median(first), " / ".join("%.3f" % statistics.median(x) for x in b), sum(statistics.median(x) for x in b)))


if __name__ == "__main__":
    main()
