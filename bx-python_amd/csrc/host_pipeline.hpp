// host_pipeline.hpp -- the host-pointer entry points' side of PCIe: page touchers for fresh output arrays, downloads behind them,
// and the chunked count (upload of chunk k+1 / pass on k / download of k-1 at once).  Host code only (threads, streams, events);
// included by intervals.hip behind bxmi_ivl_count_dev, which it calls per chunk.  bxmi_ivl_count / bxmi_ivl_find in
// include/bxmi.h say what a caller sees (the calls block; the helper threads are joined before they return).
#pragma once

// Host threads that touch the pages of an OUTPUT array chunk by chunk, ahead of the downloads into it.  A copy into fresh pageable
// memory (numpy.empty) pays for its page faults on the copying thread: 400 MB cost the download 10-16 ms, 32 ms per 100 M
// counts against 16 into touched memory.  A page is read and written back; the array is the call's output -- nobody else holds
// it -- and chunk k's download waits (wait_chunk) until its touchers are done with it, so a touch never lands on copied data.
struct PageToucher {
    char *base = nullptr;
    size_t bytes = 0, chunk = 0;
    int nchunks = 0, nthreads = 0;
    std::vector<std::atomic<int>> done;  // [chunk]: threads done with it
    std::atomic<bool> stop{false};
    std::vector<std::thread> threads;
    void start(void *p, size_t n, size_t chunk_bytes, int nthr)
    {
        base = static_cast<char *>(p), bytes = n, chunk = chunk_bytes, nthreads = nthr;
        nchunks = (int)((n + chunk_bytes - 1) / chunk_bytes);
        done = std::vector<std::atomic<int>>((size_t)nchunks);
        for (auto &d : done) d.store(0);
        for (int j = 0; j < nthreads; j++) threads.emplace_back([this, j] { run(j); });
    }
    void run(int j)
    {
        for (int k = 0; k < nchunks && !stop.load(); k++) {
            const size_t o = (size_t)k * chunk, m = std::min(chunk, bytes - o);
            volatile char *b = base + o;
            const size_t lo = m * (size_t)j / (size_t)nthreads, hi = m * (size_t)(j + 1) / (size_t)nthreads;
            for (size_t x = lo; x < hi; x += 4096) b[x] = b[x];
            if (hi > lo) b[hi - 1] = b[hi - 1];
            done[(size_t)k].fetch_add(1, std::memory_order_release);
        }
    }
    void wait_chunk(int k)
    {
        while (nthreads && done[(size_t)k].load(std::memory_order_acquire) < nthreads && !stop.load()) std::this_thread::yield();
    }
    void join()
    {
        for (auto &t : threads) t.join();
        threads.clear();
    }
    ~PageToucher()
    {
        stop.store(true);
        join();
    }
};

// Device memory -> a pageable host array the caller has not touched yet, at the link's rate: chunks of 32 MB, the touchers one
// or more chunks ahead of the copies.  `t` may be running already (started while the device was still computing); NULL = start here.
static int download_touched(void *dst, const void *src_dev, size_t bytes, hipStream_t st, PageToucher *t = nullptr)
{
    constexpr size_t CH = (size_t)32 << 20;
    if (bytes == 0) return BXMI_OK;
    PageToucher own;
    if (!t && bytes >= 2 * CH && g_opt_host_touchers > 0) own.start(dst, bytes, CH, (int)g_opt_host_touchers), t = &own;
    if (!t) {
        BXMI_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, st));
        return BXMI_OK;
    }
    for (int k = 0; k < t->nchunks; k++) {
        const size_t o = (size_t)k * t->chunk, m = std::min(t->chunk, bytes - o);
        t->wait_chunk(k);
        const hipError_t e = hipMemcpyAsync(static_cast<char *>(dst) + o, static_cast<const char *>(src_dev) + o, m, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) {
            t->stop.store(true);
            return fail(BXMI_EHIP, "download: %s", hipGetErrorString(e));
        }
    }
    return BXMI_OK;
}

// The host-pointer count in chunks: while the pass runs on chunk k (the handle's stream), chunk k+1 is on its way up (stream_up,
// this thread) and the counts of chunk k-1 on their way down (stream_down, a second host thread: a copy from or to pageable
// memory holds its caller until the runtime has staged it, and PCIe carries both directions at once only if two threads ask).
// tools/micro/pcie_probe.hip on the round's box: 56 GB/s either way alone, 47 + 47 GB/s together -- a 100 M batch is bounded by its
// 0.8 GB upload (~17 ms); one piece after the other (upload, pass, download) took 42-74 ms.
struct HostChunks {
    bxmi_ivl *h;
    int32_t *counts;
    int64_t nq, chunk;
    int nchunks;
    std::vector<hipEvent_t> done;  // chunk k's pass has finished (recorded on the handle's stream)
    std::mutex mu;
    std::condition_variable cv;
    int launched = 0;   // chunks whose pass has been launched and whose event is recorded
    std::atomic<bool> stop{false};  // the launching thread failed: nothing more will come
    int rc = BXMI_OK;
    std::string err;
    PageToucher touch;  // the output array's pages, chunk by chunk ahead of the downloads
};

static int host_chunks_download_one(HostChunks *c, int k)
{
    bxmi_ivl *h = c->h;
    const int64_t o = (int64_t)k * c->chunk, m = std::min(c->chunk, c->nq - o);
    c->touch.wait_chunk(k);
    BXMI_HIP(hipStreamWaitEvent(h->stream_down, c->done[k], 0));
    BXMI_HIP(hipMemcpyAsync(c->counts + o, h->q_cnt.as<int32_t>() + o, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream_down));
    return BXMI_OK;
}

static void host_chunks_download(HostChunks *c)
{
    if (hipSetDevice(c->h->device) != hipSuccess) {
        c->rc = BXMI_EHIP, c->err = "hipSetDevice in the download thread failed";
        return;
    }
    for (int k = 0; k < c->nchunks; k++) {
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->launched > k || c->stop; });
            if (c->launched <= k) return;
        }
        const int rc = host_chunks_download_one(c, k);
        if (rc != BXMI_OK) {
            c->rc = rc, c->err = last_error();  // (last_error() is per thread: carried over to the caller's)
            return;
        }
    }
    if (hipStreamSynchronize(c->h->stream_down) != hipSuccess) c->rc = BXMI_EHIP, c->err = "hipStreamSynchronize(stream_down) failed";
}

static int ivl_count_host_chunks(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts, int64_t *total)
{
    if (!h->stream_up) BXMI_HIP(hipStreamCreateWithFlags(&h->stream_up, hipStreamNonBlocking));
    if (!h->stream_down) BXMI_HIP(hipStreamCreateWithFlags(&h->stream_down, hipStreamNonBlocking));
    HostChunks c;
    c.h = h, c.counts = counts, c.nq = nq, c.chunk = g_opt_host_chunk, c.nchunks = (int)div_up(nq, g_opt_host_chunk);
    BXMI_TRY(h->q_s.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_e.reserve((size_t)(nq + 4) * 4));
    if (counts) BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_total.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->q_total.p, 0, 8, h->stream));
    c.done.assign((size_t)c.nchunks, nullptr);
    std::vector<hipEvent_t> up((size_t)c.nchunks, nullptr);
    auto drop_events = [&] {
        for (hipEvent_t e : c.done) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : up) if (e) (void)hipEventDestroy(e);
    };
    for (int k = 0; k < c.nchunks; k++)
        if (hipEventCreateWithFlags(&c.done[k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&up[k], hipEventDisableTiming) != hipSuccess) {
            drop_events();
            return fail(BXMI_EHIP, "bxmi_ivl_count: hipEventCreate failed");
        }
    std::thread down;
    if (counts) {
        if (g_opt_host_touchers > 0) c.touch.start(counts, (size_t)nq * 4, (size_t)c.chunk * 4, (int)g_opt_host_touchers);
        down = std::thread(host_chunks_download, &c);
    }
    auto one = [&](int k) -> int {
        const int64_t o = (int64_t)k * c.chunk, m = std::min(c.chunk, nq - o);
        BXMI_HIP(hipMemcpyAsync(h->q_s.as<int32_t>() + o, qs + o, (size_t)m * 4, hipMemcpyHostToDevice, h->stream_up));
        BXMI_HIP(hipMemcpyAsync(h->q_e.as<int32_t>() + o, qe + o, (size_t)m * 4, hipMemcpyHostToDevice, h->stream_up));
        BXMI_HIP(hipEventRecord(up[k], h->stream_up));
        BXMI_HIP(hipStreamWaitEvent(h->stream, up[k], 0));
        BXMI_TRY(bxmi_ivl_count_dev(h, h->q_s.as<int32_t>() + o, h->q_e.as<int32_t>() + o, m, counts ? h->q_cnt.as<int32_t>() + o : nullptr,
                                    h->q_total.as<int64_t>(), h->stream));  // (the chunks' totals add up in the one word)
        BXMI_HIP(hipEventRecord(c.done[k], h->stream));
        return BXMI_OK;
    };
    int rc = BXMI_OK;
    for (int k = 0; k < c.nchunks && rc == BXMI_OK; k++) {
        rc = one(k);
        std::lock_guard<std::mutex> lk(c.mu);
        if (rc == BXMI_OK) c.launched = k + 1;
        else c.stop = true, c.touch.stop.store(true);
        c.cv.notify_one();
    }
    int64_t t = 0;
    if (rc == BXMI_OK) {
        hipError_t e = hipMemcpyAsync(&t, h->q_total.p, 8, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(BXMI_EHIP, "bxmi_ivl_count: reading the total: %s", hipGetErrorString(e));
    } else
        (void)hipStreamSynchronize(h->stream);  // nothing of this call stays in flight behind its return
    if (down.joinable()) down.join();
    c.touch.join();
    drop_events();
    if (rc == BXMI_OK && c.rc != BXMI_OK) rc = fail(c.rc, "%s", c.err.c_str());
    if (rc == BXMI_OK && total) *total = t;
    return rc;
}
