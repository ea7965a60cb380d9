"""Integrators: the Python mirror of reference include/psdr/integrator/*.h, src/integrator/*.cpp.

`renderC` / `renderD` keep the reference's signatures and return conventions (src/psdr.cpp:282-294)
and run ENTIRELY on the HIP library behind include/psdr_hip.h.  There is no CPU fallback: without
libpsdr_hip.so or without a GPU the calls raise.

Multi-GPU: when torch.distributed is initialised, every rank renders the sample
slots s in [rank*spp/G, (rank+1)*spp/G) of every pixel and the image (and forward-mode
derivative image / reverse-mode gradient tables) is summed with ONE all-reduce per render call.
"""
import ctypes as C
import time

import numpy as np
import torch

import enoki as ek
from . import _abi
from .core import Object, psdr_assert, Vector3fC, Vector3fD, HyperCubeDistribution3f
from .scene import make_desc

_AD_KEYS = _abi.TANGENT_FIELDS


_SOLO = 0
_FORCE = False
_DECIDED = []          # innermost first: the collective decision a deferred render (renderD's lazy image, its backward) was created under


class solo:
    """`with integrator.solo():` -- render calls inside the block run on THIS rank alone: no spp sharding, no collective.  For work only one rank of a
    multi-rank job does while the others wait (bench.py: rank 0's counter passes and single-GPU side blocks at any world size).  A renderD issued inside
    the block keeps that decision when its image / gradient is evaluated later, outside the block (_RenderNode.collective)."""

    def __enter__(self):
        global _SOLO
        _SOLO += 1
        return self

    def __exit__(self, *exc):
        global _SOLO
        _SOLO -= 1
        return False


def force_collectives(on=True):
    """Developer switch (a function: the package reads no environment variable): the render calls issue their collectives at world size 1 too, so that every
    all-reduce of a render call EXECUTES on a one-GPU box -- over RCCL when the process group's backend is nccl (tests/test_rccl_single_rank_gpu.py; bench.py
    turns it on when ITS environment holds PSDR_FORCE_COLLECTIVES=1)."""
    global _FORCE
    _FORCE = bool(on)


class _decided:
    """Evaluate a deferred render under the collective decision taken when renderD was called."""

    def __init__(self, collective):
        self.collective = bool(collective)

    def __enter__(self):
        _DECIDED.append(self.collective)
        return self

    def __exit__(self, *exc):
        _DECIDED.pop()
        return False


def _dist():
    """torch.distributed when the render call takes part in the job's collectives: more than one rank (or force_collectives), not inside solo(), and -- for a
    deferred render -- what was decided when renderD was called."""
    import torch.distributed as dist
    if _DECIDED:
        return dist if (_DECIDED[-1] and dist.is_available() and dist.is_initialized()) else None
    if _SOLO:
        return None
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE):
        return dist
    return None


def shard_range(n, rank, world):
    """Slots [begin, end) of `n` per-pixel samples owned by `rank` (balanced, contiguous)."""
    base, rem = divmod(n, world)
    b = rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _RenderNode:
    """What renderD remembers so that a tangent (ek.forward) or an adjoint (ek.backward) can be
    pushed through the renderer later: the table tensors (with their torch graph), the render
    options including the RNG offsets of the primal call, and the guiding grid."""

    def __init__(self, integrator, scene, sensor_id, tb, opts, guide):
        self.integrator, self.scene, self.sensor_id = integrator, scene, sensor_id
        self.tb, self.opts, self.guide = tb, opts, guide
        self.collective = _dist() is not None          # the sharding of `opts` was fixed under this decision: the deferred launches keep it

    def input_tensors(self):
        return [self.tb.get(k) for k in _AD_KEYS] if self.tb is not None else [None] * len(_AD_KEYS)

    def render_forward(self, tangents):
        with _decided(self.collective):
            img, dimgs = self.integrator._render_fwd(self.scene, self.tb, self.opts, self.guide, [tangents])
        self.primal = img.reshape(-1, 3)           # by-product of the forward-mode launch: renderD's value
        return dimgs[0].reshape(-1, 3)

    def release(self):
        pass


class _LazyImage(Vector3fD):
    """What renderD returns when a scene parameter requires a gradient: the image is rendered when it is first
    LOOKED AT.  `renderD(); enoki.forward(P); enoki.gradient(img)` -- the reference harness' sequence,
    examples/run_test.py:95-142 -- then costs ONE launch (the forward-mode kernel produces image and derivative image
    together) instead of a primal launch plus a forward-mode launch; any other use (numpy(), a torch loss for
    enoki.backward) renders the primal through the autograd bridge on first access of `.t`."""

    @classmethod
    def _make(cls, node, inputs):
        o = cls.__new__(cls)
        o._node, o._inputs, o._t = node, inputs, None
        return o

    @property
    def t(self):
        if self._t is None:
            primal = getattr(self._node, "primal", None)
            if primal is not None:
                self._t = primal
            else:
                with torch.enable_grad():          # a first look under no_grad() must not cache an image without autograd history
                    self._t = _RenderFn.apply(self._node, *self._inputs)
            self._inputs = None
        return self._t

    @t.setter
    def t(self, v):
        self._t = v


class _ReducingImage(Vector3fC):
    """renderC on several GPUs: the all-reduce of the image is issued asynchronously and joined when the image is first LOOKED AT, so
    that the next render call's kernel (the harness renders renderD right after renderC) runs while the ring moves the 3 W H floats."""

    @classmethod
    def _make(cls, t, work):
        o = cls.__new__(cls)
        o._t, o._work = t, work
        return o

    @property
    def t(self):
        if self._work is not None:
            self._work.wait()                     # nccl: the current stream waits for the collective; gloo: the host does
            self._work = None
        return self._t

    @t.setter
    def t(self, v):
        self._work, self._t = None, v


class _RenderFn(torch.autograd.Function):
    """torch.autograd bridge for reverse mode: forward = primal image, backward = psdr_render_d_rev."""

    @staticmethod
    def forward(ctx, node, *inputs):
        ctx.node = node
        ctx.present = [t is not None for t in inputs]
        # this primal render is going to be differentiated in reverse mode: where the reverse launch would repeat it as its value sweep (PathTracer on a
        # two-level scene), it runs with recording stages and the backward call below runs the adjoint kernel only (PSDR_FLAG_KEEP_RECORDS)
        geo = any(t is not None and t.requires_grad for t, k in zip(inputs, _AD_KEYS) if k in ("tri_info", "cam_to_world"))
        with _decided(node.collective):
            img = node.integrator._render_c(node.scene, node.tb, node.opts, node.guide, interior_only=True, keep_records=geo)
        return img.reshape(-1, 3)

    @staticmethod
    def backward(ctx, adj):
        node = ctx.node
        with _decided(node.collective):
            grads = node.integrator._render_rev(node.scene, node.tb, node.opts, node.guide, adj.contiguous().reshape(-1))
        out = [None]
        for k, present in zip(_AD_KEYS, ctx.present):
            out.append(grads.get(k) if present else None)
        return tuple(out)


class Integrator(Object):
    """reference include/psdr/integrator/integrator.h:8-28"""
    _type_name = "SamplingIntegrator"
    _kind = _abi.INTEGRATOR_DIRECT

    def __init__(self):
        super().__init__()
        self._guide = {}
        self._counters_src, self._counters_val, self._calls = None, None, 0

    @property
    def last_counters(self):
        """(rays traced, camera / primary-edge / secondary-edge slots) of the last render call; read from the device on
        first access (a host synchronisation -- the render calls themselves do not wait for the GPU)."""
        if self._counters_val is None and self._counters_src is not None:
            lib, scene = self._counters_src
            c = (C.c_uint64 * 4)()
            lib.psdr_get_counters(scene._native, c)
            self._counters_val = tuple(int(x) for x in c)
        return self._counters_val

    # ---- option block -----------------------------------------------------------
    def _opts(self, scene, with_edges):
        o = scene.opts
        dist = _dist()
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
        sppe = o.sppe if with_edges else 0
        sppse = o.sppse if (with_edges and self._kind == _abi.INTEGRATOR_DIRECT) else 0
        opts = _abi.make_opts(
            integrator=self._kind, bsdf_samples=getattr(self, "bsdf_samples", 1), light_samples=getattr(self, "light_samples", 1),
            max_depth=getattr(self, "max_depth", 1), hide_emitters=getattr(self, "hide_emitters", False),
            field=getattr(self, "_field_id", 0), spp=o.spp, sppe=sppe, sppse=sppse,
            spp_range=shard_range(o.spp, rank, world), sppe_range=shard_range(sppe, rank, world),
            sppse_range=shard_range(sppse, rank, world), rng_offset=scene._rng_offset)
        return opts

    def _advance_rng(self, scene, opts):
        d = _abi.draws_per_slot(opts)
        if opts.spp > 0:
            scene._rng_offset[0] += d[0]
        if opts.sppe > 0:
            scene._rng_offset[1] += d[1]
        if opts.sppse > 0:
            scene._rng_offset[2] += d[2]

    # ---- native plumbing ----------------------------------------------------------
    def _prepare(self, scene, tb, guide):
        lib = _abi.load_hip()
        if not torch.cuda.is_available():
            raise RuntimeError("psdr_cuda: no GPU visible; the HIP render path has no CPU fallback")
        if scene._native is None:
            h = C.c_void_p()
            _abi.check(lib, lib.psdr_scene_create(C.byref(h)))
            scene._native = h
            for name, value in getattr(scene, "native_options", {}).items():
                _abi.check(lib, lib.psdr_scene_set_option(h, name.encode(), float(value)))
        desc, keep = make_desc(tb, guide)
        _abi.check(lib, lib.psdr_scene_set_tables(scene._native, C.byref(desc)))
        # the tree on the handle belongs to ONE set of tables: rebuild / refit whenever the tables submitted now are
        # not the ones it was built for (a renderD result differentiated after a later configure() brings its own,
        # older tables back -- and the next render call the newer ones again)
        stamp = tb.get("geo_version", scene._version)      # a material-only configure() keeps the geometry stamp
        if getattr(scene, "_bvh_version", None) != stamp:
            _abi.check(lib, lib.psdr_bvh_build(scene._native, _stream_ptr()))
            scene._bvh_version = stamp
        return lib, keep

    def _render_c(self, scene, tb, opts, guide, interior_only=False, defer_reduce=False, keep_records=False):
        lib, keep = self._prepare(scene, tb, guide)
        if keep_records and self._kind == _abi.INTEGRATOR_PATH:
            opts = type(opts).from_buffer_copy(opts)
            opts.flags |= _abi.FLAG_KEEP_RECORDS
        img = torch.empty(tb["width"] * tb["height"] * 3, dtype=torch.float32, device="cuda")
        _abi.check(lib, lib.psdr_render_c(scene._native, C.byref(opts), img.data_ptr(), _stream_ptr()))
        self._counters(lib, scene)
        dist = _dist()
        if dist and defer_reduce:
            return img, dist.all_reduce(img, async_op=True)
        if dist:
            dist.all_reduce(img)
        return (img, None) if defer_reduce else img

    def _render_fwd(self, scene, tb, opts, guide, tangent_sets):
        lib, keep = self._prepare(scene, tb, guide)
        K = len(tangent_sets)
        n = tb["width"] * tb["height"] * 3
        buf = torch.empty((1 + K) * n, dtype=torch.float32, device="cuda")   # [image || derivative images]: one all-reduce
        tarr = (_abi.Tangents * K)()
        for k, ts in enumerate(tangent_sets):
            for name, t in zip(_abi.TANGENT_FIELDS, ts):
                if t is not None:
                    t = t.detach().to(torch.float32).contiguous()
                    keep.append(t)
                    setattr(tarr[k], "d_" + name, t.data_ptr())
        _abi.check(lib, lib.psdr_render_d_fwd(scene._native, C.byref(opts), K, tarr, buf.data_ptr(),
                                              buf.data_ptr() + 4 * n, _stream_ptr()))
        self._counters(lib, scene)
        dist = _dist()
        if dist:
            dist.all_reduce(buf)
        return buf[:n], [buf[(1 + k) * n:(2 + k) * n] for k in range(K)]

    def _render_rev(self, scene, tb, opts, guide, adj):
        lib, keep = self._prepare(scene, tb, guide)
        g = _abi.Grads()
        grads, flat = {}, []
        for name in _AD_KEYS:
            t = tb.get(name)
            if t is not None and t.requires_grad:
                flat.append((name, t.shape, t.numel()))
        total = sum(n for _, _, n in flat)
        gbuf = torch.zeros(max(total, 1), dtype=torch.float32, device="cuda")   # one flat buffer: one all-reduce
        off = 0
        for name, shape, n in flat:
            grads[name] = gbuf[off:off + n].view(shape)
            setattr(g, "g_" + name, gbuf.data_ptr() + 4 * off)
            off += n
        _abi.check(lib, lib.psdr_render_d_rev(scene._native, C.byref(opts), adj.data_ptr(), None, C.byref(g), _stream_ptr()))
        self._counters(lib, scene)
        dist = _dist()
        if dist:
            dist.all_reduce(gbuf)
        return grads

    def _counters(self, lib, scene):
        """The counters stay on the device until somebody looks (last_counters): no render call reads them back (the library's
        fused-vs-wavefront choice is a function of the scene and the options, not of earlier calls)."""
        self._counters_src, self._counters_val = (lib, scene), None
        self._calls += 1

    # ---- public API (src/psdr.cpp:282-285) -------------------------------------------
    def renderC(self, scene, sensor_id=0):
        psdr_assert(scene.is_ready(), "Input scene must be configured!")
        psdr_assert(0 <= sensor_id < scene.num_sensors, "Invalid sensor id!")
        t0 = time.perf_counter()
        tb = scene.tables(sensor_id, capacity=True)
        opts = self._opts(scene, with_edges=False)
        img, work = self._render_c(scene, tb, opts, None, defer_reduce=True)
        self._advance_rng(scene, opts)
        if scene.opts.log_level:
            torch.cuda.synchronize()                     # only the log line needs the time; the image is stream-ordered
            self.log("Rendered in %g seconds." % (time.perf_counter() - t0))
        if work is not None:
            return _ReducingImage._make(img.reshape(-1, 3), work)
        return Vector3fC._wrap(img.reshape(-1, 3))

    def renderD(self, scene, sensor_id=0):
        psdr_assert(scene.is_ready(), "Input scene must be configured!")
        psdr_assert(0 <= sensor_id < scene.num_sensors, "Invalid sensor id!")
        t0 = time.perf_counter()
        tb = scene.tables(sensor_id, capacity=True)
        opts = self._opts(scene, with_edges=True)
        if not (scene._sensor_tables[sensor_id]["num_prim_edges"] > 0):
            opts.sppe = opts.sppe_begin = opts.sppe_end = 0
        guide = self._guide.get(sensor_id)
        node = _RenderNode(self, scene, sensor_id, tb, opts, guide)
        inputs = node.input_tensors()
        if any(t is not None and t.requires_grad for t in inputs):
            img = _LazyImage._make(node, inputs)           # rendered on first use (see _LazyImage)
        else:
            img = Vector3fD._wrap(self._render_c(scene, tb, opts, guide, interior_only=True).reshape(-1, 3))
            img._node = node
        self._advance_rng(scene, opts)
        ek.register_render_node(img)
        if scene.opts.log_level:
            torch.cuda.synchronize()
            self.log("Rendered in %g seconds." % (time.perf_counter() - t0))
        return img

    def preprocess_secondary_edges(self, scene, sensor_id, resolution, nrounds=1):
        """DirectIntegrator::preprocess_secondary_edges, reference direct.cpp:166-204."""
        raise RuntimeError("preprocess_secondary_edges: only DirectIntegrator builds a guiding grid")


class FieldExtractionIntegrator(Integrator):
    """reference include/psdr/integrator/field.h, src/integrator/field.cpp"""
    _type_name = "FieldExtractionIntegrator"
    _kind = _abi.INTEGRATOR_FIELD

    def __init__(self, field):
        super().__init__()
        if field not in _abi.FIELDS:
            raise RuntimeError("Unsupported field: " + str(field))
        self.m_field = field
        self._field_id = _abi.FIELDS[field]


class DirectIntegrator(Integrator):
    """reference include/psdr/integrator/direct.h, src/integrator/direct.cpp"""
    _type_name = "DirectIntegrator"
    _kind = _abi.INTEGRATOR_DIRECT

    def __init__(self, bsdf_samples=1, light_samples=1):
        super().__init__()
        psdr_assert(bsdf_samples >= 0 and light_samples >= 0 and bsdf_samples + light_samples > 0)
        self.bsdf_samples, self.light_samples = int(bsdf_samples), int(light_samples)
        self.hide_emitters = False

    def preprocess_secondary_edges(self, scene, sensor_id, resolution, nrounds=1):
        psdr_assert(nrounds > 0)
        psdr_assert(scene.is_ready(), "Scene needs to be configured!")
        reso = [int(r) for r in np.asarray(resolution).reshape(-1)]
        psdr_assert(len(reso) == 4)
        cells = reso[0] * reso[1] * reso[2]
        psdr_assert(cells * reso[3] < 2 ** 31 - 1)
        tb = scene.tables(sensor_id, capacity=True)
        lib, keep = self._prepare(scene, tb, None)
        mass = torch.zeros(cells, dtype=torch.float32, device="cuda")
        opts = self._opts(scene, with_edges=True)
        r = (C.c_int32 * 4)(*reso)
        _abi.check(lib, lib.psdr_guide_build(scene._native, C.byref(opts), r, int(nrounds), mass.data_ptr(), _stream_ptr()))
        dist = _dist()
        if dist:
            dist.all_reduce(mass)          # every rank evaluated all cells; keep replicas bit-identical
            mass /= dist.get_world_size()
        w = HyperCubeDistribution3f()
        w.set_resolution(reso[:3])
        w.set_mass(mass)
        torch.cuda.synchronize()
        self._guide[sensor_id] = (reso[:3], w.m_distrb.m_cmf, w.m_distrb.m_pmf, w.m_distrb.m_sum)
        return w


class PathTracer(Integrator):
    """Multi-bounce extension of DirectIntegrator::__Li.  NOT in the reference snapshot (SURVEY F2,
    App. F): defined so that PathTracer(max_depth=1) == DirectIntegrator(1, 1) sample for sample."""
    _type_name = "PathTracer"
    _kind = _abi.INTEGRATOR_PATH

    def __init__(self, max_depth=3):
        super().__init__()
        psdr_assert(max_depth >= 1)
        self.max_depth = int(max_depth)
        self.bsdf_samples = self.light_samples = 1
        self.hide_emitters = False
