import sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, ".")
import vp_suite_amd
from vp_suite_amd import ops
from vp_suite_amd.models import MODEL_CLASSES, ef_conv_lstm as ef
which = sys.argv[1]
x = torch.rand(4, 10, 1, 64, 64, device="cuda")
s1 = torch.cuda.Stream()
keep = []
def ev(stream):
    e = torch.cuda.Event(); e.record(stream); keep.append(e); return e
if which == "a":      # torch ops only, two streams
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        e = ev(main)
        with torch.cuda.stream(s1):
            s1.wait_event(e); y = x * 2
        main.wait_stream(s1)
    g.replay(); torch.cuda.synchronize(); print("a ok", float(y.sum()))
elif which in ("b", "c"):   # one ConvLSTM block call: b on a side stream, c on the capture stream
    W = torch.randn(256, 80, 3, 3, device="cuda") * 0.05; b = torch.zeros(256, device="cuda")
    xs = torch.rand(4, 5, 16, 64, 64, device="cuda")
    with torch.no_grad():
        for _ in range(2): ops.convlstm_seq(xs, None, None, W, b, seq_len=5, in_channels=16, precision="bf16x3")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            if which == "b":
                e = ev(main)
                with torch.cuda.stream(s1):
                    s1.wait_event(e)
                    out = ops.convlstm_seq(xs, None, None, W, b, seq_len=5, in_channels=16, precision="bf16x3")[0]
                main.wait_stream(s1)
            else:
                out = ops.convlstm_seq(xs, None, None, W, b, seq_len=5, in_channels=16, precision="bf16x3")[0]
        g.replay(); torch.cuda.synchronize(); print(which, "ok", float(out.sum()))
elif which == "d":    # whole model, ONE stream (no pipeline)
    m = MODEL_CLASSES["convlstm-shi"]("cuda", img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0., 1.], cell_precision="bf16x3").cuda()
    ef.GRAPH_SMALL_BATCH = False
    with torch.no_grad():
        for _ in range(2): m(x, pred_frames=10)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y, _ = m(x, pred_frames=10)
        g.replay(); torch.cuda.synchronize(); print("d ok", float(y.sum()))
elif which in ("e", "f", "g", "h", "i", "j", "k", "l"):
    m = MODEL_CLASSES["convlstm-shi"]("cuda", img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0., 1.], cell_precision="bf16x3").cuda()
    ef.GRAPH_SMALL_BATCH = False
    enc, fo = m.encoder, m.forecaster
    def run():
        if which == "e":   # first stage glue only
            return ef._apply_framewise(enc.stage1, x, "bf16x3", consumer=enc.rnn1)
        if which == "f":   # stage 1 + rnn1
            return enc.forward_by_stage(x, enc.stage1, enc.rnn1, enc.stage2)[0]
        if which == "g":   # encoder
            return enc(x)
        if which == "h":   # encoder + forecaster rnn3 stage
            hs = enc(x)
            return fo.forward_by_stage(None, hs[-1], 10, fo.stage3, fo.rnn3, fo.rnn2)
        if which in ("j", "k", "l"):
            hs = enc(x)
            inp = fo.forward_by_stage(None, hs[-1], 10, fo.stage3, fo.rnn3, fo.rnn2)
            inp = fo.forward_by_stage(inp, hs[1], 10, fo.stage2, fo.rnn2, fo.rnn1)
            if which == "j":
                return fo.forward_by_stage(inp, hs[0], 10, fo.stage1, fo.rnn1, None)
            if which == "k":   # rnn1 only, fp32 output
                return fo.rnn1(inp, hs[0], 10)[0]
            out, _ = fo.rnn1(inp, hs[0], 10)   # l: rnn1 (fp32 out) + glue on fp32
            return ef._apply_framewise(fo.stage1, out, "bf16x3", consumer=None)
        if which == "i":
            hs = enc(x)
            inp = fo.forward_by_stage(None, hs[-1], 10, fo.stage3, fo.rnn3, fo.rnn2)
            return fo.forward_by_stage(inp, hs[1], 10, fo.stage2, fo.rnn2, fo.rnn1)
    with torch.no_grad():
        for _ in range(2): run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = run()
        g.replay(); torch.cuda.synchronize(); print(which, "ok")
elif which in ("m", "n", "o"):
    m = MODEL_CLASSES["convlstm-shi"]("cuda", img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0., 1.], cell_precision="bf16x3").cuda()
    ef.GRAPH_SMALL_BATCH = False
    ef.PIPELINE_CHUNKS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    enc, fo = m.encoder, m.forecaster
    with torch.no_grad():
        ef._PIPE_ACTIVE = False
        hs0 = enc(x)
        ef._PIPE_ACTIVE = True
        def run():
            if which == "m":
                return enc(x)
            if which == "n":
                return fo(hs0, 10)
            return fo(enc(x), 10)
        for _ in range(2): run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = run()
        keep.extend(ef._events)
        g.replay(); torch.cuda.synchronize(); print(which, "ok")
