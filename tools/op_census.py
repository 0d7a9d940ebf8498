"""Which Python call sites issue torch ops in one eager training step?  (TorchFunctionMode: Python-level calls only;
ops created inside the C++ autograd engine -- derivative formulas, gradient accumulation -- have no Python frame.)"""
import sys; sys.path.insert(0, '.')
import collections, torch
from torch.overrides import TorchFunctionMode, resolve_name
from textboxgan_amd.config import Config
from textboxgan_amd.training_step import build_trainer_state
from bench import synthetic_batch, bench_init_
dev = torch.device('cuda:0')
cfg = Config(batch_size_per_gpu=16)
st = build_trainer_state(cfg, dev, seed=0, use_graphs=False); bench_init_(st)
b = synthetic_batch(cfg, dev, 1234); ts = st["training_step"]; 
args = (b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], False, False, 1e-4)
for _ in range(2): ts.dist_train_step(*args)
torch.cuda.synchronize()
SKIP = {"Tensor.shape.__get__", "Tensor.device.__get__", "Tensor.dtype.__get__", "Tensor.is_cuda.__get__", "Tensor.dim", "Tensor.size",
        "Tensor.numel", "Tensor.data_ptr", "Tensor.is_contiguous", "Tensor.requires_grad.__get__", "Tensor.is_leaf.__get__",
        "Tensor._version.__get__", "Tensor.stride", "Tensor.view", "Tensor.reshape", "Tensor.expand", "Tensor.transpose",
        "Tensor.permute", "Tensor.detach", "Tensor.squeeze", "Tensor.unsqueeze", "Tensor.__getitem__", "Tensor.t", "Tensor.chunk",
        "Tensor.grad_fn.__get__", "Tensor.record_stream", "Tensor.element_size", "Tensor.storage_offset", "Tensor.untyped_storage"}
cnt = collections.Counter()
class Census(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = resolve_name(func) or getattr(func, "__name__", str(func))
        name = name.replace("torch.", "")
        if name not in SKIP:
            f = sys._getframe(1); site = "?"
            while f is not None:
                fn = f.f_code.co_filename
                if "textboxgan_amd" in fn:
                    site = f"{fn.split('/')[-1]}:{f.f_lineno} {f.f_code.co_name}"; break
                f = f.f_back
            cnt[(name, site)] += 1
        return func(*args, **(kwargs or {}))
with Census():
    ts.dist_train_step(*args)
torch.cuda.synchronize()
print("total python-level torch calls (excl. views/metadata):", sum(cnt.values()))
for (name, site), n in cnt.most_common(int(sys.argv[1]) if len(sys.argv) > 1 else 70):
    print(f"{n:5d}  {name:32s} {site}")
