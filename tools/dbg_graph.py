import sys,time,torch,faulthandler
faulthandler.enable()
sys.path.insert(0,".")
import vp_suite_amd
from vp_suite_amd.models import MODEL_CLASSES, ef_conv_lstm as ef
ef.PIPELINE_CHUNKS=int(sys.argv[1]); ef.GRAPH_SMALL_BATCH = sys.argv[2]=="1"
if len(sys.argv)>3: ef._PIPE_ACTIVE = True
m=MODEL_CLASSES["convlstm-shi"]("cuda",img_shape=(1,64,64),action_size=0,tensor_value_range=[0.,1.],cell_precision="bf16x3").cuda()
x=torch.rand(4,10,1,64,64,device="cuda")
with torch.no_grad():
    print("eager first", flush=True)
    y,_=m(x,pred_frames=10); torch.cuda.synchronize(); print("ok", float(y.sum()), flush=True)
    for _ in range(5): m(x,pred_frames=10)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(100): m(x,pred_frames=10)
    torch.cuda.synchronize(); print("ms/step",(time.perf_counter()-t0)*10, flush=True)
