// polar_multi.cpp — lifetime of the multi-device context (polar_multi.h) and of the per-device copies of a handle.
#include "polar_multi.h"

namespace polar_host {

Rccl g_rccl;
std::atomic<int> g_comm_inits{0};

void multi_release(polar_code *h, bool abort_comms) {
    MultiCtx *m = h->multi;
    if (!m) return;
    h->multi = nullptr;
    if (m->stuck) {
        // a worker never came back from the driver: nothing it may still touch is freed (MultiCtx::run_all step 3)
        for (auto &t : m->threads) t.detach();
        h->multi_poisoned = true;
        return;
    }
    m->stop_workers();
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t d = 0; d < m->comms.size(); ++d)
        if (void *c = m->take_comm((int)d)) {
            // after a failed round a rank may be stuck inside a collective: abort, do not wait for it
            if (abort_comms && g_rccl.CommAbort) (void)g_rccl.CommAbort(c);
            else (void)g_rccl.CommDestroy(c);
        }
    for (size_t d = 0; d < m->streams.size(); ++d)
        if (m->streams[d]) { (void)hipSetDevice(m->devs[d]); (void)hipStreamDestroy(m->streams[d]); }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete m;
}
// the handle's tables on another device (owned by `h`, reused by later calls)
polar_code *clone_on_device(polar_code *h, int dev, bool fresh) {
    // fresh: a context of its own even when one exists for this device (test hook share_device)
    if (!fresh) {
        if (dev == h->device) return h;
        for (polar_code *c : h->clones) if (c->device == dev) return c;
    }
    polar_code *c = copy_ctx(h, dev);
    h->clones.push_back(c);
    return c;
}
// a copy of the handle's tables and settings bound to `dev`, with its own (not yet allocated) device state; the caller owns it
polar_code *copy_ctx(polar_code *h, int dev) {
    polar_code *c = new polar_code;
    c->n = h->n; c->N = h->N; c->K = h->K; c->crc = h->crc; c->eps = h->eps;
    c->frozen = h->frozen; c->order = h->order; c->bitrev = h->bitrev; c->crcm = h->crcm;
    c->W = h->W; c->info_rank = h->info_rank; c->crc_mask = h->crc_mask; c->sched = h->sched; c->ctl = h->ctl;
    c->sc_ops = h->sc_ops; c->sc_lat_ops = h->sc_lat_ops; c->sc_fold = h->sc_fold; c->weak_leaves = h->weak_leaves;
    c->device = dev;
    c->waves_per_cu = h->waves_per_cu; c->lds_log = h->lds_log; c->pipe = h->pipe; c->prefix_on = h->prefix_on; c->mode = h->mode;
    c->knobs = h->knobs;
    return c;
}

}  // namespace polar_host
