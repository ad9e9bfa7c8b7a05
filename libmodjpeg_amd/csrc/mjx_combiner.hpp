// mjx_combiner.hpp — combining concurrent callers of a host-buffer entry point into one batch.
//
// The entropy coders are chains of small, latency-bound kernels that occupy a sliver of the GPU, so
// N concurrent per-image calls cost N times the chain when each launches its own.  Here every caller
// stages its own image (own arena, own stream, an event recorded after its uploads) and joins a
// queue; the first one to find a free batch slot becomes the leader and launches ONE set of kernels
// over everything queued at that moment (the kernels take an array of per-image descriptors and use
// blockIdx.y as the image), the others sleep until their image is done.  A lone caller is a batch
// of one and waits for nobody, so single-threaded latency is unchanged.
#pragma once
#include "mjx_internal.hpp"

#include <pthread.h>

namespace mjx {

// R needs: Arena *arena; bool taken, done;
template <class R, int MAX_BATCH> struct Combiner {
    pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t  finished = PTHREAD_COND_INITIALIZER;
    R              *queue[256]; // at most one request per staging arena
    int             queued = 0, leaders = 0;

    // run(lead, batch, n) launches the batch on lead->arena->stream, waits for it and fills in every
    // request's result; max_leaders = batches that may be in flight at once
    template <class RUN> void submit(R *req, int max_leaders, RUN &&run) {
        pthread_mutex_lock(&lock);
        queue[queued++] = req;
        while(!req->done) {
            if(req->taken || leaders >= max_leaders) { // somebody else's batch holds this image, or enough batches are in flight
                pthread_cond_wait(&finished, &lock);
                continue;
            }
            R   *batch[MAX_BATCH];
            int  n = 0;
            bool mine = false;
            for(int i = 0; i < queued && n < MAX_BATCH; ++i) {
                if(queue[i]->arena->device != req->arena->device) continue; // a batch is one launch set on one GPU
                mine |= queue[i] == req;
                batch[n++] = queue[i];
            }
            if(!mine) batch[n - 1] = req; // more queued than a batch takes: the leader's own image must be in it
            int left = 0;
            for(int i = 0; i < queued; ++i) {
                bool in_batch = false;
                for(int k = 0; k < n; ++k) in_batch |= batch[k] == queue[i];
                if(!in_batch) queue[left++] = queue[i];
            }
            queued = left;
            for(int k = 0; k < n; ++k) batch[k]->taken = true;
            leaders++;
            pthread_mutex_unlock(&lock);
            run(req, batch, n);
            pthread_mutex_lock(&lock);
            for(int k = 0; k < n; ++k) batch[k]->done = true;
            leaders--;
            pthread_cond_broadcast(&finished);
        }
        pthread_mutex_unlock(&lock);
    }
};

} // namespace mjx
