# shapes around the bench default: the reference's batchSizeGPU = 8, odd sizes, a large batch; then 300 steps for drift / leaks
set -e
for b in 8 5 33 128 200; do
  python bench.py --cpu-seconds 0 --no-prof --steps 5 --warmup 2 --batch $b 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch', d['config']['windows_per_gpu'], d['ms_per_step'], 'ms', 'loss', d['config']['final_losses'][0])"
done
python - <<'PY'
import torch, bench, time
from cpc2_amd.train import DataParallelContext, cpcStep
cfg = bench.CONFIGS["small"]
dev = torch.device("cuda:0")
model, crit, opt = bench.build(cfg, dev)
dp = DataParallelContext(opt)
crit.seed(1)
g = torch.Generator().manual_seed(0)
x = (0.05 * torch.randn(16, 1, bench.WINDOW, generator=g)).to(dev)
label = torch.zeros(16, dtype=torch.long, device=dev)
first = last = None
torch.cuda.reset_peak_memory_stats()
for step in range(300):
    tot, losses, acc = cpcStep(x, x, label, model, crit)
    tot.backward(); dp.reduce_and_step(); opt.zero_grad()
    if step == 10: mem10 = torch.cuda.memory_allocated()
    if step == 0: first = losses.mean().item()
last = losses.mean().item()
print("loss", first, "->", last, "acc", acc.mean().item(), "allocated MB step10/300", mem10 >> 20, torch.cuda.memory_allocated() >> 20)
assert last < first and torch.isfinite(losses).all()
assert torch.cuda.memory_allocated() <= mem10 * 1.02 + (1 << 20)
PY
