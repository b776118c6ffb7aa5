"""Kernel sequence of ONE inference step (convlstm-shi, 64x64, 10 -> 10; BB = batch, default 4), for the question "where does a small-batch
step spend its time": run on the GPU box as
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b4trace -o b4 -- python3 tools/trace_step.py run
    python3 tools/trace_step.py show > gpurun_out/b4_step.txt
`run` does 40 forwards; `show` prints the last complete step of the trace: kernel, duration, gap to its predecessor
(profiles/r06_b4_step_trace.txt)."""
import sys, os, csv, glob


def run():
    sys.path.insert(0, os.getcwd())
    import torch
    from vp_suite_amd.models import MODEL_CLASSES
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B = int(os.environ.get("BB", 4))
    model = MODEL_CLASSES["convlstm-shi"](str(dev), img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0], cell_precision="bf16x3").to(dev)
    x = torch.rand(B, 10, 1, 64, 64, device=dev)
    with torch.no_grad():
        for _ in range(40):
            model(x, pred_frames=10)
    torch.cuda.synchronize()


def show():
    f = glob.glob("gpurun_out/b4trace/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "conv_few_to_16" in r["Kernel_Name"]]   # the first launch of a step
    a, b = idx[-2], idx[-1]
    prev_end = None
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print("%-72s dur %7.1f us gap %6.1f us" % (r["Kernel_Name"][:72], (e - s) / 1e3, gap))
        prev_end = e
    print("step span us", (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)


if __name__ == "__main__":
    (run if sys.argv[1:] == ["run"] else show)()
