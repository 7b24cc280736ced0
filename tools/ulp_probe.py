import torch
g = torch.Generator().manual_seed(0)
a = torch.randn(1<<20, generator=g); b = torch.randn(1<<20, generator=g)*3+0.1; c = torch.randn(1<<20, generator=g)
ad,bd,cd = a.cuda(), b.cuda(), c.cuda()
def cmp(name, x_cpu, x_gpu):
    d = (x_cpu.view(torch.int32) != x_gpu.cpu().view(torch.int32)).sum().item()
    print(name, "mismatching elements:", d)
cmp("div", a/b, ad/bd)
cmp("mul", a*b, ad*bd)
cmp("mul_then_add", a*b+c, ad*bd+cd)
cmp("sub_div", (a-b)/(c-b), (ad-bd)/(cd-bd))
cmp("exp", torch.exp(a), torch.exp(ad))
cmp("recip", 1.0/b, 1.0/bd)
n=12
mid=(torch.arange(n).float()+0.5)
near=torch.rand(1000,1); far=near+torch.rand(1000,1)*3
t_c = mid/n*(far-near)+near
t_g = mid.cuda()/n*(far.cuda()-near.cuda())+near.cuda()
cmp("sample_depth", t_c, t_g)
