"""What the data-parallel launch form pays on ONE GPU, piece by piece (no process group; bench configuration c2, batch 2):
  single : the whole step as one hipGraph
  V0     : TrainStep(data_parallel=True).run() -- 5 graphs, per-pass hand-over to the communication stream, AdamW per piece
  V1     : the same 5 graphs replayed back to back, AdamW for every piece afterwards on the SAME stream (no second stream)
  V2     : V1 + an (empty) hand-over after every graph: comm_stream.wait_stream(main)
  V3     : V1 with one torch.cuda.Event().record() on the main stream after every graph"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
from bench import CFG  # noqa: E402
from tools.synthetic import synthetic_volume  # noqa: E402

dev = torch.device("cuda:0")


def build(dp):
    torch.manual_seed(1234)
    model = pkg.UNETRLogits(**CFG).to(dev)
    model.precision = "bf16"
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    flat = model.use_flat_buffers()
    opt = pkg.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, flat=flat)
    x, y = synthetic_volume(2, 1, 96, 4, seed=1234)
    return pkg.TrainStep(model, crit, opt, x.to(dev), y.to(dev), data_parallel=dp, handover=os.environ.get("PROBE_HANDOVER", "stream"))


def timed(fn, steps=20, windows=3):
    for _ in range(5):
        fn()
    res = []
    for _ in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e3)
    return sorted(res)[len(res) // 2]


def main():
    global st, all_pieces, hip, flag, one, count, Fn
    st = build(False)
    only = os.environ.get("PROBE_ONLY", "single,V0,V1,V2,V3,V4,V5,V6").split(",")
    if "single" in only:
        print(f"single : {timed(st.run):.3f} ms/step")
    del st
    st = build(True)
    if "V0" in only:
        print(f"V0     : {timed(st.run):.3f} ms/step")
    all_pieces = [p for ps in st.pieces for p in ps]


    def update_all():
        steps = st.opt.begin_reduced_step(st._plan)
        for lo, hi in all_pieces:
            st.opt.step_runs(st._plan, st._runs_of[(lo, hi)], steps, st.flat["grad"], 1.0)
        st.opt.end_reduced_step(st._plan)


    def v1():
        for g in st.graphs:
            g.replay()
        update_all()


    def v2():
        for g in st.graphs:
            g.replay()
            st.comm_stream.wait_stream(torch.cuda.current_stream())
        update_all()


    def v3():
        for g in st.graphs:
            g.replay()
            torch.cuda.Event().record()
        update_all()


    if "V1" in only:
        print(f"V1     : {timed(v1):.3f} ms/step")
    if "V2" in only:
        print(f"V2     : {timed(v2):.3f} ms/step")
    if "V3" in only:
        print(f"V3     : {timed(v3):.3f} ms/step")

    # ---- the hand-over as a stream memory operation: a kernel behind every graph bumps a float counter in signal memory, the
    # communication stream waits for the value (hipStreamWaitValue32); the main stream records / waits nothing
    import ctypes  # noqa: E402
    import struct  # noqa: E402
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
    flag = ctypes.c_void_p()
    assert hip.hipExtMallocWithFlags(ctypes.byref(flag), ctypes.c_size_t(8), ctypes.c_uint(0x2)) == 0
    hip.hipMemset(flag, 0, ctypes.c_size_t(8))
    torch.cuda.synchronize()
    one = torch.ones(1, device=dev)
    count = [0]
    Fn = pkg.functional


    def bits(f):
        return struct.unpack("<I", struct.pack("<f", f))[0]


    def signal_and_wait():
        count[0] += 1
        Fn.call("unetr_counter_add", flag.value, one.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
        assert hip.hipStreamWaitValue32(ctypes.c_void_p(st.comm_stream.cuda_stream), flag, bits(float(count[0])), 0, 0xFFFFFFFF) == 0


    def v4():
        for g in st.graphs:
            g.replay()
            signal_and_wait()
        update_all()


    def v5():
        main = torch.cuda.current_stream()
        steps = None
        for g, ks in zip(st.graphs, st.graph_passes):
            g.replay()
            k = ks[-1]
            signal_and_wait()
            with torch.cuda.stream(st.comm_stream):
                if steps is None:
                    steps = st.opt.begin_reduced_step(st._plan)
                for lo, hi in st.pieces[k]:
                    st.opt.step_runs(st._plan, st._runs_of[(lo, hi)], steps, st.flat["grad"], 1.0)
        with torch.cuda.stream(st.comm_stream):
            st.opt.end_reduced_step(st._plan)
        main.wait_stream(st.comm_stream)


    def v6():
        # hand-over through the HOST: an event behind every graph; the launching thread runs one graph ahead, waits for the event
        # of the pass before (the GPU is busy with the next graph meanwhile) and only then issues that pass's communication-stream
        # work -- the communication stream never waits on an unfinished event of the main stream
        main = torch.cuda.current_stream()
        state = {"steps": None}

        def comm_work(ev, k):
            ev.synchronize()
            with torch.cuda.stream(st.comm_stream):
                if state["steps"] is None:
                    state["steps"] = st.opt.begin_reduced_step(st._plan)
                for lo, hi in st.pieces[k]:
                    st.opt.step_runs(st._plan, st._runs_of[(lo, hi)], state["steps"], st.flat["grad"], 1.0)
        pending = None
        for g, ks in zip(st.graphs, st.graph_passes):
            g.replay()
            ev = torch.cuda.Event()
            ev.record(main)
            if pending is not None:
                comm_work(*pending)
            pending = (ev, ks[-1])
        comm_work(*pending)
        with torch.cuda.stream(st.comm_stream):
            st.opt.end_reduced_step(st._plan)
        main.wait_stream(st.comm_stream)

    if "V6" in only:
        print(f"V6     : {timed(v6):.3f} ms/step   (graphs + event per graph, hand-over through the host, AdamW per piece on the communication stream)")
    if "V4" in only:
        print(f"V4     : {timed(v4):.3f} ms/step   (5 graphs, flag + wait-value per graph, AdamW afterwards on the main stream)")
    if "V5" in only:
        print(f"V5     : {timed(v5):.3f} ms/step   (5 graphs, flag + wait-value per pass, AdamW per piece on the communication stream)")
    if "V1" in only:
        print(f"V1     : {timed(v1):.3f} ms/step")


if __name__ == "__main__":
    main()
