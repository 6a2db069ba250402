import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
import torch, bench
class D:  # minimal dist
    local_rank=0; rank=0; world=1
    dev=torch.device('cuda:0')
    def barrier(self): pass
    def max_over_ranks(self,x): return x
torch.cuda.set_device(0)
bank = bench.ReceiverBank(0, D.dev, 128, 1<<22, 0, True)
for _ in range(10): bank.step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(40): bank.step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:7000])
