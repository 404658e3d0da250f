#!/usr/bin/env python3
"""Device-side hand-over of the result records to torch, as every rank of the multi-GPU bench does it: records copied
device to device into a torch tensor on the context's own stream, that stream wrapped as a torch ExternalStream and
ordered against the consumer (torch's current stream) with events.  torch is imported BEFORE libconga_hip.so is
loaded (it ships its own HIP runtime; the other order leaves the library without a device)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conga_amd import capi, synth  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    c = synth.make_chrom("13", 2_000_000, cov=1.0, n_dels=100, n_dups=0)
    ds, de = synth.kept_sorted(c.del_start, c.del_end)
    with capi.Context(device=0, flags=capi.FLAG_BATCH | capi.FLAG_RESULTS_ON_DEVICE) as ctx:
        ctx.chrom_begin(c.length, c.gc)
        ctx.reads(c.pos, c.mapq)
        ctx.intervals("D", ds, de)
        ext = torch.cuda.ExternalStream(ctx.stream(), device=dev)
        n_bytes = len(ds) * capi.RESULT_DTYPE.itemsize
        buf = torch.zeros(n_bytes + 128, dtype=torch.uint8, device=dev)
        ev, taken = None, None
        for _ in range(4):
            if ev is not None:
                ext.wait_event(ev)  # the consumer of the previous round is done with `buf`
            ctx.compute()
            ctx.results_copy(buf.data_ptr(), n_bytes)
            ctx.sync()
            taken = buf.clone()  # stands in for the collective on torch's current stream
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        rec = np.frombuffer(taken[:n_bytes].cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
        dels = ctx.fetch()[0]
        assert rec.tobytes() == dels.tobytes(), "device records differ from the fetched ones"
    print("ok: %d records through an ExternalStream" % len(ds))


if __name__ == "__main__":
    main()
