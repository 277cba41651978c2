import sys, time, math, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi
torch.manual_seed(0)
B=int(sys.argv[1]) if len(sys.argv)>1 else 32
which=sys.argv[2] if len(sys.argv)>2 else 'fwd,bwd'
C=O=256; K=9
H=int(os.environ.get('QB_H',56)); W=int(os.environ.get('QB_W',56))
dev='cuda'
x=torch.randn(B,C,H,W,device=dev); off=torch.randn(B,18,H,W,device=dev); m=torch.sigmoid(torch.randn(B,9,H,W,device=dev))
w=(torch.rand(O,C,3,3,device=dev)*2-1)/math.sqrt(C*K); b=torch.randn(O,device=dev)*0.1; go=torch.randn(B,O,H,W,device=dev)
geo=(3,3,1,1,1,1,1,1,1,1,64,True)
def fwd(): return M.modulated_deform_conv2d_forward_cuda(x,w,b,off,m,*geo)
def bwd(): return M.modulated_deform_conv2d_backward_cuda(x,w,b,off,m,go,*geo)
for name,fn in (('fwd',fwd),('bwd',bwd)):
    if name not in which: continue
    fn(); torch.cuda.synchronize()
    n=10 if name=='fwd' else 3
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); dt=e0.elapsed_time(e1)/n
    fl=2*B*O*C*K*H*W*(1 if name=='fwd' else 2)
    print(name, 'HxW %dx%d tiles %d'%(H,W,B*H*W//32), 'ms %.3f'%dt, 'TFLOP/s %.1f'%(fl/dt*1e-9), 'path', _capi.last_path())
