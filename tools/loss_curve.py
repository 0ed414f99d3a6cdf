"""Loss / grad-norm per update of the bench workload (sanity of the training dynamics): python tools/loss_curve.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


class A:
    arch = "s2t_transformer_m"; ctc_layer = 8; batch = int(os.environ.get("B", 64)); frames = 1500; tgt_len = 40; cpu_baseline = False; lr = float(os.environ.get("LR", "5e-3")); attn_2d = bool(int(os.environ.get("ATTN2D", "0")))


dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
a, task, model, crit, trainer, _ = bench.build_all(A, dev, torch.bfloat16)
sample = trainer.prepare(task.dummy_batch(seed=100))
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    trainer.train_step([sample])
    st = trainer.reduce_stats()
    lens = model.encoder._last["lengths_host"]
    print("update %2d loss/sample %9.3f gnorm %9.2f  frames after CTC compression: max %d mean %.1f (of %d)" % (
        i + 1, st["loss"] / max(st["sample_size"], 1), st.get("gnorm", float("nan")), max(lens), sum(lens) / len(lens), A.frames // 4))
