"""Timestep streaming: upload of timestep t+1 overlapped with build / levelCut / ray-cast of
timestep t on separate HIP streams with double-buffered device volumes.
The reference does these strictly one after another (main.cpp:242-290): load -> build ->
levelCut -> glTexImage3D -> draw.  torch is plumbing here (pinned host memory, streams, events);
all compute is libvrhip.so."""
import torch

from .codec import BrickSet


class TimestepStreamer:
    def __init__(self, num_bricks, brick_dims, tolerance=1, max_epochs=2):
        self.bs = BrickSet(num_bricks, brick_dims, tolerance, max_epochs)
        n = num_bricks * brick_dims[0] * brick_dims[1] * brick_dims[2]
        self.vox = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(2)]
        self.out = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(2)]
        self.copy_stream = torch.cuda.Stream()
        self.compute_stream = torch.cuda.Stream()

    def run(self, host_timesteps, on_decoded=None, overlap=True):
        """host_timesteps: list of pinned uint8 host tensors (one per timestep).  on_decoded(t, volume, stream)
        is called with the decoded device volume while `stream` is current (launch the frame there).
        Returns the per-timestep tree infos."""
        infos = []
        uploaded = [torch.cuda.Event() for _ in host_timesteps]
        consumed = [torch.cuda.Event() for _ in host_timesteps]
        T = len(host_timesteps)

        def upload(t):
            with torch.cuda.stream(self.copy_stream):
                if t >= 2:
                    self.copy_stream.wait_event(consumed[t - 2])      # the buffer's previous user has finished
                self.vox[t & 1].copy_(host_timesteps[t], non_blocking=True)
                uploaded[t].record(self.copy_stream)

        upload(0)
        for t in range(T):
            if overlap and t + 1 < T:
                upload(t + 1)
            with torch.cuda.stream(self.compute_stream):
                self.compute_stream.wait_event(uploaded[t])
                self.bs.build(self.vox[t & 1], stream=self.compute_stream)
                self.bs.decode(self.out[t & 1], stream=self.compute_stream)
                if on_decoded is not None:
                    on_decoded(t, self.out[t & 1], self.compute_stream)
                consumed[t].record(self.compute_stream)
            # per-timestep public members (synchronises the compute stream, like the reference's prints)
            infos.append([self.bs.info(b) for b in range(self.bs.num_bricks)])
            if not overlap and t + 1 < T:
                upload(t + 1)
                self.copy_stream.synchronize()
        torch.cuda.synchronize()
        return infos
