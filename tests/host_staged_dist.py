"""TEST INFRASTRUCTURE: a torch.distributed look-alike whose point-to-point ops stage device tensors through the host, so that two
ranks of the product's tiling code (mega-minecraft_amd/distributed.py: TileContext, exchange_placements, the overlap with region_fill)
can run as two PROCESSES on ONE GPU over a gloo group (RCCL refuses two ranks on one device; gloo cannot send device tensors).
Everything else - the ring kernels, the two-phase protocol, the order of cells on the wire - is the product's own code."""


class _Req:
    def __init__(self, work, copy_back):
        self.work, self.copy_back = work, copy_back

    def wait(self):
        self.work.wait()
        if self.copy_back is not None:
            dst, src = self.copy_back
            dst.copy_(src)


class HostStagedDist:
    isend, irecv = "isend", "irecv"

    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch
        self._keep = []

    def P2POp(self, op, tensor, peer):
        return (op, tensor, peer)

    def batch_isend_irecv(self, ops):
        reqs = []
        for op, t, peer in ops:
            if op == self.isend:
                cpu = t.cpu().contiguous()
                self._keep.append(cpu)
                reqs.append(_Req(self.dist.isend(cpu, peer), None))
            else:
                cpu = self.torch.empty(t.shape, dtype=t.dtype)
                reqs.append(_Req(self.dist.irecv(cpu, peer), (t, cpu)))
        return reqs
