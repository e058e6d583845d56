/*
 * wire -- the few host-side exchanges a multi-GPU run of sph_run needs, over TCP, with rank 0
 * as the hub: a broadcast (the RCCL id), the per-frame gather of the ranks' rows for the .plt
 * file, and -- for ranks that share a GPU, where RCCL cannot form a communicator -- the
 * host-staged transport callbacks of include/sph_amd.h (sendrecv with the slab neighbours,
 * all-reduce).  No counterpart in the reference, which is one process.
 *
 * Every operation is collective: all ranks call it at the same point.  Particle data of a real
 * multi-GPU run never goes through here (it moves over RCCL inside the library); what does is
 * small or per frame.
 */
#ifndef SPH_WIRE_H
#define SPH_WIRE_H

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

class Wire {
public:
    int rank = 0, size = 1;
    std::string error;
    /* No message of a run is larger than this (the driver sets it from the capacity of its
     * context): a length beyond it read from a peer ends the exchange instead of sizing a buffer. */
    uint64_t max_message = 1ull << 32;
    /* seconds a receive may wait for its peer before the exchange fails (a rank that died or
     * never started must not leave the others blocked for ever) */
    int recv_timeout_s = 600;

    /* closes every connection: peers blocked in an exchange with this rank then fail instead of
     * waiting (called by the driver before it exits on an error) */
    void shutdown_all() {
        for (int& fd : fds_)
            if (fd >= 0) {
                ::shutdown(fd, SHUT_RDWR);
                close(fd);
                fd = -1;
            }
        if (listen_fd_ >= 0) {
            close(listen_fd_);
            listen_fd_ = -1;
        }
    }

    ~Wire() {
        for (int fd : fds_)
            if (fd >= 0) close(fd);
        if (listen_fd_ >= 0) close(listen_fd_);
    }

    /* rank 0 listens on addr:port (the address the ranks were given: loopback unless the launcher
     * names another) and accepts size-1 peers within timeout_s; the others connect (retrying
     * while rank 0 is not up yet) and say who they are */
    bool open(int rank_, int size_, const std::string& addr, int port, int timeout_s = 120) {
        rank = rank_;
        size = size_;
        if (size <= 1) return true;
        addrinfo hints, *res = nullptr;
        memset(&hints, 0, sizeof hints);
        hints.ai_family = AF_INET;
        hints.ai_socktype = SOCK_STREAM;
        const std::string ps = std::to_string(port);
        if (getaddrinfo(addr.c_str(), ps.c_str(), &hints, &res) != 0 || !res) return fail("getaddrinfo");
        const auto t0 = std::chrono::steady_clock::now();
        if (rank == 0) {
            listen_fd_ = socket(AF_INET, SOCK_STREAM, 0);
            if (listen_fd_ < 0) { freeaddrinfo(res); return fail("socket"); }
            int one = 1;
            setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
            const int bound = bind(listen_fd_, res->ai_addr, res->ai_addrlen);
            freeaddrinfo(res);
            if (bound != 0) return fail("bind");
            if (listen(listen_fd_, size) != 0) return fail("listen");
            fds_.assign((size_t)size, -1);
            for (int k = 1; k < size; ++k) {
                /* a peer that never starts must not hold rank 0 for ever */
                const long left = timeout_s - (long)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count();
                timeval tv;
                tv.tv_sec = left > 1 ? left : 1;
                tv.tv_usec = 0;
                setsockopt(listen_fd_, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
                const int fd = accept(listen_fd_, nullptr, nullptr);
                if (fd < 0) return fail("accept (a rank did not connect in time)");
                setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
                set_timeouts(fd);
                int32_t who = -1;
                if (!recv_all(fd, &who, sizeof who) || who < 1 || who >= size || fds_[(size_t)who] >= 0) {
                    /* not one of this run's ranks: drop it and keep waiting for them */
                    close(fd);
                    --k;
                    continue;
                }
                fds_[(size_t)who] = fd;
            }
            return true;
        }
        int fd = -1;
        for (;;) {
            fd = socket(AF_INET, SOCK_STREAM, 0);
            if (fd < 0) break;
            if (connect(fd, res->ai_addr, res->ai_addrlen) == 0) break;
            close(fd);
            fd = -1;
            if (std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count() > timeout_s) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        freeaddrinfo(res);
        if (fd < 0) return fail("connect to rank 0");
        int one = 1;
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        set_timeouts(fd);
        const int32_t who = rank;
        if (!send_all(fd, &who, sizeof who)) return fail("hello");
        fds_.assign(1, fd);
        return true;
    }

    /* the bytes of rank 0 to everybody */
    bool bcast(void* buf, size_t n) {
        if (size <= 1) return true;
        if (rank == 0) {
            for (int k = 1; k < size; ++k)
                if (!send_all(fds_[(size_t)k], buf, n)) return fail("bcast send");
            return true;
        }
        return recv_all(fds_[0], buf, n) || fail("bcast recv");
    }

    /* in place; op 0 = max, 1 = sum; folded in rank order on rank 0, so every rank gets the same bits */
    bool allreduce(double* v, int n, int op) {
        if (size <= 1) return true;
        if (rank != 0) return (send_all(fds_[0], v, sizeof(double) * (size_t)n) && recv_all(fds_[0], v, sizeof(double) * (size_t)n)) || fail("allreduce");
        std::vector<double> in((size_t)n);
        for (int k = 1; k < size; ++k) {
            if (!recv_all(fds_[(size_t)k], in.data(), sizeof(double) * (size_t)n)) return fail("allreduce recv");
            for (int i = 0; i < n; ++i) v[i] = op ? v[i] + in[(size_t)i] : (in[(size_t)i] > v[i] ? in[(size_t)i] : v[i]);
        }
        for (int k = 1; k < size; ++k)
            if (!send_all(fds_[(size_t)k], v, sizeof(double) * (size_t)n)) return fail("allreduce send");
        return true;
    }

    /* each rank hands nsL bytes to rank-1 and nsR bytes to rank+1 and gets what they hand to it
     * (relayed by rank 0) */
    bool sendrecv(const void* sL, uint64_t nsL, void* rL, uint64_t nrL, const void* sR, uint64_t nsR, void* rR, uint64_t nrR) {
        if (size <= 1) return true;
        if (rank != 0) {
            const uint64_t head[2] = {nsL, nsR};
            const int fd = fds_[0];
            if (!send_all(fd, head, sizeof head) || !send_all(fd, sL, nsL) || !send_all(fd, sR, nsR)) return fail("sendrecv send");
            uint64_t got[2];
            if (!recv_all(fd, got, sizeof got) || got[0] != nrL || got[1] != nrR) return fail("sendrecv sizes");
            return (recv_all(fd, rL, nrL) && recv_all(fd, rR, nrR)) || fail("sendrecv recv");
        }
        std::vector<std::vector<char> > toL((size_t)size), toR((size_t)size); /* what rank k hands to its left / right */
        toL[0].assign((const char*)sL, (const char*)sL + nsL);
        toR[0].assign((const char*)sR, (const char*)sR + nsR);
        for (int k = 1; k < size; ++k) {
            uint64_t head[2];
            if (!recv_all(fds_[(size_t)k], head, sizeof head)) return fail("sendrecv head");
            if (head[0] > max_message || head[1] > max_message) return fail("sendrecv: a peer names a length beyond any message of this run");
            toL[(size_t)k].resize(head[0]);
            toR[(size_t)k].resize(head[1]);
            if (!recv_all(fds_[(size_t)k], toL[(size_t)k].data(), head[0]) || !recv_all(fds_[(size_t)k], toR[(size_t)k].data(), head[1]))
                return fail("sendrecv body");
        }
        for (int k = 0; k < size; ++k) {
            static const std::vector<char> none;
            const std::vector<char>& fromL = k > 0 ? toR[(size_t)k - 1] : none;        /* left neighbour's right-going bytes */
            const std::vector<char>& fromR = k + 1 < size ? toL[(size_t)k + 1] : none; /* right neighbour's left-going bytes */
            if (k == 0) {
                if (fromL.size() != nrL || fromR.size() != nrR) return fail("sendrecv sizes");
                if (nrR) memcpy(rR, fromR.data(), nrR);
                continue;
            }
            const uint64_t got[2] = {fromL.size(), fromR.size()};
            const int fd = fds_[(size_t)k];
            if (!send_all(fd, got, sizeof got) || !send_all(fd, fromL.data(), fromL.size()) || !send_all(fd, fromR.data(), fromR.size()))
                return fail("sendrecv relay");
        }
        return true;
    }

    /* rank 0 receives every other rank's buffer (sizes may differ); others send theirs */
    bool gather(const void* mine, uint64_t n, std::vector<std::vector<char> >* all) {
        if (size <= 1) return true;
        if (rank != 0) return (send_all(fds_[0], &n, sizeof n) && send_all(fds_[0], mine, n)) || fail("gather send");
        all->assign((size_t)size, std::vector<char>());
        for (int k = 1; k < size; ++k) {
            uint64_t m = 0;
            if (!recv_all(fds_[(size_t)k], &m, sizeof m)) return fail("gather head");
            if (m > max_message) return fail("gather: a peer names a length beyond any message of this run");
            (*all)[(size_t)k].resize(m);
            if (!recv_all(fds_[(size_t)k], (*all)[(size_t)k].data(), m)) return fail("gather body");
        }
        return true;
    }

private:
    std::vector<int> fds_;
    int listen_fd_ = -1;

    void set_timeouts(int fd) const {
        timeval tv;
        tv.tv_sec = recv_timeout_s;
        tv.tv_usec = 0;
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
    }

    bool fail(const char* what) {
        error = std::string("wire: ") + what + (errno ? std::string(": ") + strerror(errno) : std::string());
        return false;
    }
    static bool send_all(int fd, const void* p, uint64_t n) {
        const char* c = (const char*)p;
        while (n) {
            const ssize_t k = ::send(fd, c, (size_t)std::min<uint64_t>(n, 1u << 30), MSG_NOSIGNAL);
            if (k <= 0) {
                if (k < 0 && errno == EINTR) continue;
                return false;
            }
            c += k;
            n -= (uint64_t)k;
        }
        return true;
    }
    static bool recv_all(int fd, void* p, uint64_t n) {
        char* c = (char*)p;
        while (n) {
            const ssize_t k = ::recv(fd, c, (size_t)std::min<uint64_t>(n, 1u << 30), 0);
            if (k <= 0) {
                if (k < 0 && errno == EINTR) continue;
                return false;
            }
            c += k;
            n -= (uint64_t)k;
        }
        return true;
    }
};

#endif
