"""G15: GEM on an AlexNet-structured net — the reference's UNCHANGED methods/rehearsal/model/gem.py (dev container only).

gem.Net.forward's manual dropout (gem.py:166-196), observe (gem.py:206-287) and observe_FT (gem.py:288-309) on the small
AlexNet-structured net of g15_inputs.py:

  * step 0, 1: observe on task 0 (no memory pass yet): masks drawn per observe, loss, hits;
  * step 2, 3: observe on task 1: memory pass over task 0's exemplars + current batch share ONE mask set; Gram test and
    (if violated) projection; loss, hits, projected count;
  * then init_setup + two observe_FT steps (masks drawn once, kept);
  * parameters after the observe steps and after the FT steps; the eval-mode logits of the final net.

Harness pieces (not reference code): exemplars are 'paths' = indices into the fixture's own batch bank, and the exemplar
image folder / data loader of common.RehearsalMemory are replaced by one in-order batch from that bank (no image files
here); quadprog = tests/golden/harness/quadprog.py.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import harness  # noqa: E402

torch = harness.install()
import g15_inputs as I  # noqa: E402
import methods.rehearsal.model.gem as GEM  # noqa: E402

LR, WD, MARGIN = 0.002, 0.0, 0.5


def main():
    torch.manual_seed(int(os.environ.get("G15_SEED", "3")))     # head init and mask draws (both recorded in the fixture)
    tmp = tempfile.mkdtemp()
    base = I.load_params(I.SmallAlexNet(), I.det_params())
    path = os.path.join(tmp, "base.pth.tar")
    torch.save(base, path)
    args = types.SimpleNamespace(prev_model_path=path, cuda=False, n_memories=I.N_MEM, nc_per_task=I.NC_PER_TASK, lr=LR,
                                 weight_decay=WD, memory_strength=MARGIN, batch_size=I.BATCH,
                                 task_imgfolders={"train": types.SimpleNamespace(transform=None)})
    net = GEM.Net(0, I.N_OUT, 2, args)
    bank = {}
    data = I.batches(steps=6)
    out = {}
    # the extended head (rows 4..7) is torch-default-initialised by gem.Net.__init__: carry it in the fixture
    for i, p in enumerate(net.parameters()):
        out["p0_%d" % i] = p.detach().numpy().copy()

    def patch_memory():
        md = net.memory_data
        md.get_imagefolder = lambda exemplarlist, targetlist, transform: (exemplarlist, targetlist)
        md.get_dataloader = lambda folder, batch_size=None: iter(
            [(torch.stack([bank[k] for k in folder[0]]), folder[1].clone())])

    for step in range(4):
        t = 0 if step < 2 else 1
        x, y = (torch.from_numpy(a) for a in data[step])
        keys = [(step, i) for i in range(len(y))]
        for k, xi in zip(keys, x):
            bank[k] = xi
        if net.memory_data is not None:
            patch_memory()
        loss, hits, stats = net.observe(x, t, y, keys, args)
        patch_memory()
        out["s%d_loss" % step] = loss.detach().numpy().copy()
        out["s%d_hits" % step] = np.array(int(hits))
        out["s%d_proj" % step] = np.array(stats["projected_grads"][0])
        for idx, m in net.dropout_masks.items():
            out["s%d_mask%d" % (step, idx)] = m.numpy().copy()
        out["s%d_mem_labels" % step] = net.memory_labels.numpy().copy()
        if step == 3:
            out["grads_cols"] = net.grads.numpy().copy()
    for i, p in enumerate(net.parameters()):
        out["p4_%d" % i] = p.detach().numpy().copy()
    net.init_setup(args)
    net.train(True)
    for step in (4, 5):
        x, y = (torch.from_numpy(a) for a in data[step])
        loss, hits = net.observe_FT(x, 1, y)
        out["s%d_loss" % step] = loss.detach().numpy().copy()
        out["s%d_hits" % step] = np.array(int(hits))
        for idx, m in net.dropout_masks.items():
            out["s%d_mask%d" % (step, idx)] = m.numpy().copy()
    for i, p in enumerate(net.parameters()):
        out["p6_%d" % i] = p.detach().numpy().copy()
    net.eval()
    with torch.no_grad():
        out["eval_logits_t1"] = net.forward(torch.from_numpy(data[0][0]), 1).numpy().copy()
    np.savez_compressed(os.path.join(HERE, "G15_gem_alexnet.npz"), **out)
    print({k: (v.shape if v.ndim else v.item()) for k, v in out.items() if not k.startswith("p")})


if __name__ == "__main__":
    main()
