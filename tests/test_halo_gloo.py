"""world_size-2 (and 3) halo exchange on CPU over gloo: every rank builds its topology and
halo plan through the C ABI (cup3d_grid_create_uniform / cup3d_grid_halo_plan), packs its face
slabs in the documented slab layout (a numpy restatement of k_pack_faces), exchanges one
message per peer in plan order with torch.distributed, and checks every received slab against
the ghost values read directly from the global field.  This is the N>1 data path of
halo_exchange() (comm.hip) with gloo send/recv standing in for ncclSend/ncclRecv."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def pack_slab(block, f, nc, w):
    """block: [8,8,8,nc] (z,y,x,c) -> slab [nc, w, 64] with lane = a2*8 + a1 (advdiff.hip k_pack_faces)."""
    d, side = f >> 1, f & 1
    out = np.empty((nc, w, 64))
    for gl in range(w):
        q = 7 - gl if side else gl
        if d == 2:
            plane = block[q, :, :, :]              # (a2=y, a1=x)
        elif d == 1:
            plane = block[:, q, :, :]              # (a2=z, a1=x)
        else:
            plane = block[:, :, q, :]              # (a2=z, a1=y)
        out[:, gl, :] = plane.reshape(64, nc).T
    return out


def worker(rank, world, port, bpd, level, bc, nc, w, q):
    import cup3d_amd as cu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = cu.Grid(bpd, level + 1, level, 1.0, bc, rank, world)
        N = [(b << level) * 8 for b in bpd]
        rng = np.random.default_rng(7)
        glob = rng.uniform(-1, 1, (N[2], N[1], N[0], nc))     # same on every rank
        mine = g.to_blocks(glob)
        send, recv, faces = g.halo_plan()
        sendbuf = np.concatenate([pack_slab(mine[sf // 6], sf % 6, nc, w).ravel() for sf in faces]) if len(faces) else np.zeros(0)
        recvbuf = np.zeros(int(recv.sum()) * nc * w * 64)
        per = nc * w * 64
        reqs, so, ro = [], 0, 0
        st, rt = torch.from_numpy(sendbuf), torch.from_numpy(recvbuf)
        for p in range(world):                                  # the grouped Send/Recv loop of halo_exchange()
            ns, nr = int(send[p]) * per, int(recv[p]) * per
            if ns:
                reqs.append(dist.isend(st[so:so + ns], p))
            if nr:
                reqs.append(dist.irecv(rt[ro:ro + nr], p))
            so += ns
            ro += nr
        for r in reqs:
            r.wait()
        slabs = recvbuf.reshape(-1, nc, w, 64)
        nbr = g.neighbours()
        checked = 0
        for s in range(g.nblocks):
            for f in range(6):
                n = int(nbr[s, f])
                if n < cu.capi.NBR_HALO:
                    continue
                d, side = f >> 1, f & 1
                base = g.index[s] * 8
                for gl in range(w):
                    for a2 in range(8):
                        for a1 in range(8):
                            c = [0, 0, 0]
                            c[d] = 8 + gl if side else -1 - gl
                            t = [x for x in range(3) if x != d]
                            c[t[0]], c[t[1]] = a1, a2
                            gx, gy, gz = [(int(base[k]) + c[k]) % N[k] for k in range(3)]
                            assert np.array_equal(slabs[n - cu.capi.NBR_HALO][:, gl, a2 * 8 + a1], glob[gz, gy, gx]), (rank, s, f, gl)
                checked += 1
        assert checked == int(recv.sum())
        q.put((rank, "ok", checked))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "fail", repr(e)))
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,bpd,level,bc,nc,w", [
    (2, (2, 2, 2), 1, (1, 1, 1), 3, 3),     # periodic, velocity 3-deep (advect-diffuse)
    (2, (2, 1, 1), 1, (2, 1, 0), 1, 1),     # mixed BCs, scalar 1-deep (Poisson LHS)
    (3, (1, 1, 1), 2, (1, 2, 1), 3, 1),     # uneven partition, velocity 1-deep (pressure RHS)
])
def test_halo_exchange_over_gloo(world, bpd, level, bc, nc, w):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, bpd, level, bc, nc, w, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
    assert sum(r[2] for r in res) > 0
