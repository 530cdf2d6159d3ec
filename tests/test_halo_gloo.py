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


# ------------------------------------------------------------------ multi-level meshes: ghost blocks + face fluxes of Grid.rank_view
def amr_worker(rank, world, port, golden, q):
    import cup3d_amd as cu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(golden)
        t = g["tables"]
        mesh = cu.Grid(tuple(int(b) for b in g["bpd"]), int(g["level_max"]), 0, 1.0, tuple(int(b) for b in g["bc"]),
                       leaves=(t[:, 0].astype(np.int32), t[:, 1].copy()))
        nb = mesh.nblocks
        owner = (np.arange(nb) * world // nb).astype(np.int32)            # contiguous runs of the global order
        v = mesh.rank_view(owner, rank, world)
        faces, _, _ = mesh.interface()
        rng = np.random.default_rng(11)
        field = rng.uniform(-1, 1, (nb, 3, 512))                          # a global field and global face fluxes, same on every rank
        flux = rng.uniform(-1, 1, (len(faces), 3, 64))
        mine = np.zeros((v.nlocal + v.nghost, 3, 512))
        mine[:v.nlocal] = field[v.global_slot[:v.nlocal]]                 # a rank holds its own blocks ...
        myflux = np.zeros((len(v.faces), 3, 64))
        myflux[:v.nfaces_local] = flux[v.global_face[:v.nfaces_local]]    # ... and computes the fluxes of its own faces
        for data, send_list, scount, rcount, first in ((mine, v.send_blocks, v.send_block_count, v.recv_block_count, v.nlocal),
                                                        (myflux, v.send_flux_faces, v.send_flux_count, v.recv_flux_count, v.nfaces_local)):
            per = data[0].size
            sendbuf = torch.from_numpy(np.ascontiguousarray(data[send_list]).reshape(-1)) if len(send_list) else torch.zeros(0, dtype=torch.float64)
            recv_t = torch.from_numpy(data[first:].reshape(-1))           # ghosts are received in place, one contiguous run per peer
            reqs, so, ro = [], 0, 0
            for p in range(world):
                ns, nr = int(scount[p]) * per, int(rcount[p]) * per
                if ns:
                    reqs.append(dist.isend(sendbuf[so:so + ns], p))
                if nr:
                    reqs.append(dist.irecv(recv_t[ro:ro + nr], p))
                so += ns
                ro += nr
            for r in reqs:
                r.wait()
        assert np.array_equal(mine, field[v.global_slot]) and np.array_equal(myflux, flux[v.global_face])
        # ... and the SUB-BOX form of the ghost-block exchange (what comm.hip ships by default: k_pack_boxes / k_unpack_boxes restated with
        # the plan of cup3d_grid_view_boxes): of every sent block the cells of its box, [c][z][y][x], blocks in the order of send_blocks;
        # they arrive in the boxes of the ghost blocks, in slot order.  Both width classes; the cells outside the boxes stay NaN.
        saved = 0
        for k in (0, 1):
            nc = 3

            def cells(block, box):
                x0, y0, z0, x1, y1, z1 = (int(t) for t in box)
                return block.reshape(nc, 8, 8, 8)[:, z0:z1, y0:y1, x0:x1]

            parts = [cells(mine[sl], bx).reshape(-1) for sl, bx in zip(v.send_blocks, v.send_box[k])]
            sendbuf = torch.from_numpy(np.concatenate(parts)) if parts else torch.zeros(0, dtype=torch.float64)
            assert sendbuf.numel() == int(v.send_cells[k].sum()) * nc
            recvbuf = torch.zeros(int(v.recv_cells[k].sum()) * nc, dtype=torch.float64)
            reqs, so, ro = [], 0, 0
            for p in range(world):
                ns, nr = int(v.send_cells[k][p]) * nc, int(v.recv_cells[k][p]) * nc
                if ns:
                    reqs.append(dist.isend(sendbuf[so:so + ns], p))
                if nr:
                    reqs.append(dist.irecv(recvbuf[ro:ro + nr], p))
                so += ns
                ro += nr
            for r in reqs:
                r.wait()
            got = np.full((v.nghost, nc, 8, 8, 8), np.nan)
            rb, o = recvbuf.numpy(), 0
            for i, bx in enumerate(v.ghost_box[k]):
                x0, y0, z0, x1, y1, z1 = (int(t) for t in bx)
                n = nc * (x1 - x0) * (y1 - y0) * (z1 - z0)
                got[i, :, z0:z1, y0:y1, x0:x1] = rb[o:o + n].reshape(nc, z1 - z0, y1 - y0, x1 - x0)
                o += n
            assert o == rb.size
            want = field[v.global_slot[v.nlocal:]].reshape(v.nghost, nc, 8, 8, 8)
            shipped = ~np.isnan(got)
            assert np.array_equal(got[shipped], want[shipped])
            assert shipped.sum() == rb.size, (k, int(shipped.sum()), rb.size)
            saved += got.size - int(shipped.sum())
        q.put((rank, "ok", v.nghost + v.nfaces_ghost + saved))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", repr(e) + traceback.format_exc()[-600:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "amr_mixed_l12"), (3, "amr_periodic_l01")])
def test_ghost_block_and_flux_exchange_over_gloo(world, name, golden_dir):
    """The two exchanges a multi-level mesh spread over ranks needs (whole ghost blocks before a stencil kernel, face-flux arrays
    after a flux-corrected one), carried out with the plans of cup3d_grid_rank_view over gloo: afterwards every visible slot /
    interface face of a rank holds the data of the global mesh.  Then the ghost-block exchange in its SUB-BOX form (the library's default),
    both stencil-width classes: what arrives in the ghost blocks' boxes is the global mesh's data, cell for cell, and nothing else travels."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=amr_worker, args=(r, world, port, os.path.join(golden_dir, name + ".npz"), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
    assert sum(r[2] for r in res) > 0
