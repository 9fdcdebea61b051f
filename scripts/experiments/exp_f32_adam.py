import sys
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'cobel-rl_amd'))
import torch, bench
from cobel_amd.network import TorchNetwork
n_in=int(sys.argv[1]); n,B,gamma,tau=23,32,0.8,0.01
def build(dtname, fused):
    torch.manual_seed(3)
    proto=TorchNetwork(bench._mlp(n_in,4,'f32'), optimizer_params={'lr':2e-3,'weight_decay':0.0})
    proto.set_device(torch.device('cuda',0))
    net=proto.replicate(n)
    gen=torch.Generator(device='cuda').manual_seed(7)
    with torch.no_grad():
        for p in net.params.values():
            p.add_(0.05*torch.randn(p.shape,generator=gen,device='cuda',dtype=torch.float32))
    net.fused_mlp=fused
    return net
fused, plain = build('f32',True), build('f32',False)
ft, pt = fused.clone(), plain.clone()
with torch.no_grad():
    for a,b in zip(ft.params.values(), pt.params.values()): a.mul_(0.9); b.mul_(0.9)
# float64 twin of plain
import copy
P64={k:v.detach().double().clone() for k,v in plain.params.items()}
T64={k:v.detach().double().clone() for k,v in pt.params.items()}
gen=torch.Generator(device='cuda').manual_seed(11)
s=torch.rand((n,B,n_in),generator=gen,device='cuda')*2-0.5
ns=torch.rand((n,B,n_in),generator=gen,device='cuda')*2-0.5
a=torch.randint(0,4,(n,B),generator=gen,device='cuda')
r=torch.rand((n,B),generator=gen,device='cuda'); nt=(torch.rand((n,B),generator=gen,device='cuda')<0.8).float()
assert fused.dqn_replay_fused(ft,s,a,r,ns,nt,gamma,True,tau,None)
with torch.no_grad():
    targets=plain.forward(s).clone(); boot=pt.forward(ns); pick=plain.forward(ns).argmax(dim=2)
    boot=torch.gather(boot,2,pick[...,None])[...,0]
    targets.scatter_(2,a[...,None],(r+boot*nt*gamma)[...,None])
plain.train_on_device(s,targets,None,blend_into=pt,tau=tau)
# float64 truth: manual forward/backward with autograd
names=list(P64)
W={k:v.clone().requires_grad_(True) for k,v in P64.items()}
def fwd(Wd,x):
    h=torch.relu(torch.einsum('nbi,nhi->nbh',x,Wd['dense_1.weight'])+Wd['dense_1.bias'][:,None])
    h=torch.relu(torch.einsum('nbi,nhi->nbh',h,Wd['dense_2.weight'])+Wd['dense_2.bias'][:,None])
    return torch.einsum('nbi,nhi->nbh',h,Wd['output.weight'])+Wd['output.bias'][:,None]
s6,ns6,r6,nt6=s.double(),ns.double(),r.double(),nt.double()
with torch.no_grad():
    t6=fwd(P64,s6).clone(); b6=fwd(T64,ns6); pk=fwd(P64,ns6).argmax(dim=2)
    b6=torch.gather(b6,2,pk[...,None])[...,0]; t6.scatter_(2,a[...,None],(r6+b6*nt6*gamma)[...,None])
out=fwd(W,s6); loss=((out-t6)**2).mean(dim=(1,2)).sum(); loss.backward()
lr,eps=2e-3,1e-8
for k in ['dense_2.weight','output.bias','dense_1.weight']:
    g=W[k].grad; upd=P64[k]-lr*g/(g.abs()+eps)   # Adam step 1: m_hat/ (sqrt(v_hat)+eps) = g/(|g|+eps)
    ef=(fused.params[k].double()-upd).abs().max().item(); ep=(plain.params[k].double()-upd).abs().max().item()
    d=(fused.params[k]-plain.params[k]).abs().max().item()
    small=(g.abs()<1e-6).sum().item()
    print(k,'fused-vs-f64 %.3e torch-vs-f64 %.3e fused-vs-torch %.3e ; |g|<1e-6: %d of %d'%(ef,ep,d,small,g.numel()))
