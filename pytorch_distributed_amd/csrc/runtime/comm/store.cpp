// TCP key-value store used to bootstrap communicators (see store.h).
//
// Wire format (little endian, one request at a time per connection):
//   handshake  client -> "PDASTOR1" u32 len, job id      server -> u8 1 (accepted) or closes
//   SET  u8 1, u32 klen, key, u32 vlen, value             -> u8 1
//   GET  u8 2, u32 klen, key                              -> u32 vlen, value   (parked until set)
//   ADD  u8 3, u32 klen, key, i64 delta                   -> i64 new value
//   DEL  u8 4, u32 klen, prefix                           -> u8 1   (every key starting with prefix)
#include "store.h"

#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstring>
#include <stdexcept>

namespace pda {
namespace comm {

namespace {
constexpr char kMagic[8] = {'P', 'D', 'A', 'S', 'T', 'O', 'R', '1'};
enum Op : uint8_t { kSet = 1, kGet = 2, kAdd = 3, kDel = 4 };

double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

void put_u32(std::string& s, uint32_t v) { s.append(reinterpret_cast<const char*>(&v), 4); }
void put_str(std::string& s, const std::string& v) {
  put_u32(s, (uint32_t)v.size());
  s.append(v);
}

bool resolve(const std::string& host, int port, sockaddr_storage* out, socklen_t* len) {
  addrinfo hints{};
  hints.ai_family = AF_UNSPEC;
  hints.ai_socktype = SOCK_STREAM;
  addrinfo* res = nullptr;
  const std::string p = std::to_string(port);
  if (getaddrinfo(host.empty() ? nullptr : host.c_str(), p.c_str(), &hints, &res) != 0 || res == nullptr) return false;
  std::memcpy(out, res->ai_addr, res->ai_addrlen);
  *len = res->ai_addrlen;
  freeaddrinfo(res);
  return true;
}
}  // namespace

namespace net {

void set_nodelay(int fd) {
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
}

int listen_on(const std::string& host, int port, int backlog, std::string* bound_host) {
  // First choice: the one address `host` names, so the port is not open on interfaces nobody was
  // told about. When the name does not resolve to an IPv4 address of this machine (a virtual IP,
  // a name that only peers resolve) the bind fails with EADDRNOTAVAIL and every interface is used.
  if (bound_host) bound_host->clear();
  for (int attempt = host.empty() ? 1 : 0; attempt < 2; ++attempt) {
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_addr.s_addr = htonl(INADDR_ANY);
    addr.sin_port = htons((uint16_t)port);
    if (attempt == 0) {
      sockaddr_storage ss{};
      socklen_t len = 0;
      if (!resolve(host, port, &ss, &len) || ss.ss_family != AF_INET) continue;
      addr.sin_addr = reinterpret_cast<sockaddr_in*>(&ss)->sin_addr;
    }
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) throw std::runtime_error("store: socket() failed");
    int one = 1;
    setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    if (::bind(fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) == 0 && ::listen(fd, backlog) == 0) {
      if (attempt == 0 && bound_host) {
        char buf[INET_ADDRSTRLEN] = {0};
        inet_ntop(AF_INET, &addr.sin_addr, buf, sizeof(buf));
        *bound_host = buf;
      }
      return fd;
    }
    const int err = errno;
    ::close(fd);
    if (attempt == 0 && err == EADDRINUSE) return -1;  // the port is taken on that address: next port
  }
  return -1;
}

int bound_port(int fd) {
  sockaddr_in addr{};
  socklen_t len = sizeof(addr);
  getsockname(fd, reinterpret_cast<sockaddr*>(&addr), &len);
  return ntohs(addr.sin_port);
}

int connect_to(const std::string& host, int port, double timeout_s) {
  const double deadline = now_s() + timeout_s;
  do {
    sockaddr_storage ss{};
    socklen_t len = 0;
    if (resolve(host, port, &ss, &len)) {
      int fd = ::socket(ss.ss_family, SOCK_STREAM, 0);
      if (fd >= 0) {
        if (::connect(fd, reinterpret_cast<sockaddr*>(&ss), len) == 0) {
          set_nodelay(fd);
          return fd;
        }
        ::close(fd);
      }
    }
    ::usleep(20 * 1000);
  } while (now_s() < deadline);
  return -1;
}

void send_all(int fd, const void* data, size_t n) {
  const char* p = static_cast<const char*>(data);
  while (n > 0) {
    ssize_t w = ::send(fd, p, n, MSG_NOSIGNAL);
    if (w < 0) {
      if (errno == EINTR) continue;
      if (errno == EAGAIN || errno == EWOULDBLOCK) {
        pollfd pf{fd, POLLOUT, 0};
        ::poll(&pf, 1, 1000);
        continue;
      }
      throw std::runtime_error(std::string("comm: send failed: ") + std::strerror(errno));
    }
    p += w;
    n -= (size_t)w;
  }
}

void recv_all(int fd, void* data, size_t n, double timeout_s) {
  char* p = static_cast<char*>(data);
  const double deadline = now_s() + timeout_s;
  while (n > 0) {
    pollfd pf{fd, POLLIN, 0};
    const double left = deadline - now_s();
    if (left <= 0) throw std::runtime_error("comm: timed out waiting for a peer (is another rank dead or stuck?)");
    int pr = ::poll(&pf, 1, (int)std::min(left * 1000.0 + 1.0, 1000.0));
    if (pr < 0 && errno != EINTR) throw std::runtime_error("comm: poll failed");
    if (pr <= 0) continue;
    ssize_t r = ::recv(fd, p, n, 0);
    if (r == 0) throw std::runtime_error("comm: peer closed the connection");
    if (r < 0) {
      if (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK) continue;
      throw std::runtime_error(std::string("comm: recv failed: ") + std::strerror(errno));
    }
    p += r;
    n -= (size_t)r;
  }
}

std::string local_address(int fd) {
  sockaddr_storage ss{};
  socklen_t len = sizeof(ss);
  if (getsockname(fd, reinterpret_cast<sockaddr*>(&ss), &len) != 0) return "127.0.0.1";
  char buf[INET6_ADDRSTRLEN] = {0};
  if (ss.ss_family == AF_INET) {
    inet_ntop(AF_INET, &reinterpret_cast<sockaddr_in*>(&ss)->sin_addr, buf, sizeof(buf));
  } else {
    inet_ntop(AF_INET6, &reinterpret_cast<sockaddr_in6*>(&ss)->sin6_addr, buf, sizeof(buf));
  }
  return buf;
}

}  // namespace net

// ---------------------------------------------------------------------------------- server
StoreServer::StoreServer(const std::string& host, int base_port, int scan, std::string job_id)
    : job_id_(std::move(job_id)) {
  for (int i = 0; i < scan && listen_fd_ < 0; ++i) {
    listen_fd_ = net::listen_on(host, base_port + i, 256, &bound_host_);
    if (listen_fd_ >= 0) port_ = base_port + i;
  }
  if (listen_fd_ < 0)
    throw std::runtime_error("store: no free port in [" + std::to_string(base_port) + ", " +
                             std::to_string(base_port + scan) + ")");
  if (::pipe(wake_fd_) != 0) throw std::runtime_error("store: pipe() failed");
  thread_ = std::thread([this] { loop(); });
}

StoreServer::~StoreServer() {
  // Ranks leave a job at slightly different times: a peer may still be waiting for the reply to its
  // last barrier when rank 0 is already shutting down. Keep serving until every other client has
  // closed its connection (bounded, so a crashed peer cannot hold rank 0 for ever).
  for (int i = 0; i < 100 && open_clients_.load() > 0; ++i) ::usleep(100 * 1000);
  stop_ = true;
  char b = 1;
  (void)!::write(wake_fd_[1], &b, 1);
  if (thread_.joinable()) thread_.join();
  for (auto& kv : clients_) ::close(kv.first);
  ::close(listen_fd_);
  ::close(wake_fd_[0]);
  ::close(wake_fd_[1]);
}

size_t StoreServer::num_keys() {
  std::lock_guard<std::mutex> lock(data_mu_);
  return data_.size();
}

void StoreServer::answer_waiters(const std::string& key) {
  auto it = waiters_.find(key);
  if (it == waiters_.end()) return;
  const std::string& v = data_[key];
  std::string reply;
  put_str(reply, v);
  for (int fd : it->second) {
    try {
      net::send_all(fd, reply.data(), reply.size());
    } catch (const std::exception&) {
    }
  }
  waiters_.erase(it);
}

bool StoreServer::consume(Client& c) {
  std::lock_guard<std::mutex> data_lock(data_mu_);
  auto& in = c.in;
  size_t pos = 0;
  auto need = [&](size_t n) { return in.size() - pos >= n; };
  auto u32_at = [&](size_t at) {
    uint32_t v;
    std::memcpy(&v, in.data() + at, 4);
    return v;
  };
  for (;;) {
    const size_t start = pos;
    if (!c.greeted) {
      if (!need(12)) break;
      if (std::memcmp(in.data() + pos, kMagic, 8) != 0) return false;
      const uint32_t n = u32_at(pos + 8);
      if (n > 4096) return false;
      if (!need(12 + n)) break;
      const std::string job(reinterpret_cast<const char*>(in.data() + pos + 12), n);
      if (job != job_id_) return false;  // another job's client found this port: turn it away
      pos += 12 + n;
      c.greeted = true;
      const uint8_t ok = 1;
      net::send_all(c.fd, &ok, 1);
      continue;
    }
    if (!need(5)) break;
    const uint8_t op = in[pos];
    const uint32_t klen = u32_at(pos + 1);
    if (klen > (1u << 20)) return false;
    if (!need(5 + klen)) break;
    const std::string key(reinterpret_cast<const char*>(in.data() + pos + 5), klen);
    size_t p = pos + 5 + klen;
    if (op == kSet) {
      if (in.size() - p < 4) break;
      const uint32_t vlen = u32_at(p);
      if (in.size() - p < 4 + (size_t)vlen) break;
      data_[key] = std::string(reinterpret_cast<const char*>(in.data() + p + 4), vlen);
      pos = p + 4 + vlen;
      const uint8_t ok = 1;
      net::send_all(c.fd, &ok, 1);
      answer_waiters(key);
    } else if (op == kGet) {
      pos = p;
      auto it = data_.find(key);
      if (it != data_.end()) {
        std::string reply;
        put_str(reply, it->second);
        net::send_all(c.fd, reply.data(), reply.size());
      } else {
        waiters_[key].push_back(c.fd);
      }
    } else if (op == kAdd) {
      if (in.size() - p < 8) break;
      int64_t delta;
      std::memcpy(&delta, in.data() + p, 8);
      pos = p + 8;
      int64_t cur = 0;
      auto it = data_.find(key);
      if (it != data_.end() && it->second.size() == 8) std::memcpy(&cur, it->second.data(), 8);
      cur += delta;
      data_[key] = std::string(reinterpret_cast<const char*>(&cur), 8);
      net::send_all(c.fd, &cur, 8);
      answer_waiters(key);
    } else if (op == kDel) {
      pos = p;
      for (auto it = data_.lower_bound(key); it != data_.end() && it->first.compare(0, key.size(), key) == 0;)
        it = data_.erase(it);
      const uint8_t ok = 1;
      net::send_all(c.fd, &ok, 1);
    } else {
      return false;
    }
    (void)start;
  }
  in.erase(in.begin(), in.begin() + (long)pos);
  return true;
}

void StoreServer::loop() {
  std::vector<pollfd> fds;
  while (!stop_) {
    fds.clear();
    fds.push_back({listen_fd_, POLLIN, 0});
    fds.push_back({wake_fd_[0], POLLIN, 0});
    for (auto& kv : clients_) fds.push_back({kv.first, POLLIN, 0});
    if (::poll(fds.data(), fds.size(), 500) <= 0) continue;
    if (fds[0].revents & POLLIN) {
      int fd = ::accept(listen_fd_, nullptr, nullptr);
      if (fd >= 0) {
        net::set_nodelay(fd);
        Client c;
        c.fd = fd;
        clients_[fd] = std::move(c);
        ++open_clients_;
      }
    }
    for (size_t i = 2; i < fds.size(); ++i) {
      if (!(fds[i].revents & (POLLIN | POLLHUP | POLLERR))) continue;
      const int fd = fds[i].fd;
      auto it = clients_.find(fd);
      if (it == clients_.end()) continue;
      uint8_t buf[65536];
      ssize_t r = ::recv(fd, buf, sizeof(buf), 0);
      bool keep = r > 0;
      if (keep) {
        it->second.in.insert(it->second.in.end(), buf, buf + r);
        try {
          keep = consume(it->second);
        } catch (const std::exception&) {
          keep = false;
        }
      }
      if (!keep) {
        for (auto& w : waiters_) {
          auto& v = w.second;
          for (size_t k = 0; k < v.size();) {
            if (v[k] == fd) v.erase(v.begin() + (long)k);
            else ++k;
          }
        }
        ::close(fd);
        clients_.erase(it);
        --open_clients_;
      }
    }
  }
}

// ---------------------------------------------------------------------------------- client
StoreClient::StoreClient(const std::string& host, int base_port, int scan, const std::string& job_id, double timeout_s)
    : timeout_s_(timeout_s) {
  const double deadline = now_s() + timeout_s;
  std::string hello(kMagic, 8);
  put_str(hello, job_id);
  while (fd_ < 0 && now_s() < deadline) {
    for (int i = 0; i < scan && fd_ < 0; ++i) {
      int fd = net::connect_to(host, base_port + i, 0.0);
      if (fd < 0) continue;
      try {
        net::send_all(fd, hello.data(), hello.size());
        uint8_t ok = 0;
        net::recv_all(fd, &ok, 1, 5.0);
        if (ok == 1) {
          fd_ = fd;
          break;
        }
      } catch (const std::exception&) {
      }
      ::close(fd);
    }
    if (fd_ < 0) ::usleep(50 * 1000);
  }
  if (fd_ < 0)
    throw std::runtime_error("store: could not reach the rendezvous server at " + host + ":" + std::to_string(base_port) +
                             " within " + std::to_string((int)timeout_s) + " s (is rank 0 running?)");
}

StoreClient::~StoreClient() {
  if (fd_ >= 0) ::close(fd_);
}

void StoreClient::set(const std::string& key, const std::string& value) {
  std::lock_guard<std::mutex> lock(mu_);
  std::string req(1, (char)kSet);
  put_str(req, key);
  put_str(req, value);
  net::send_all(fd_, req.data(), req.size());
  uint8_t ok = 0;
  net::recv_all(fd_, &ok, 1, timeout_s_);
}

std::string StoreClient::get(const std::string& key) {
  std::lock_guard<std::mutex> lock(mu_);
  std::string req(1, (char)kGet);
  put_str(req, key);
  net::send_all(fd_, req.data(), req.size());
  uint32_t n = 0;
  try {
    net::recv_all(fd_, &n, 4, timeout_s_);
  } catch (const std::exception& e) {
    throw std::runtime_error(std::string(e.what()) + " [store key '" + key + "']");
  }
  std::string v(n, '\0');
  if (n) net::recv_all(fd_, v.data(), n, timeout_s_);
  return v;
}

int64_t StoreClient::add(const std::string& key, int64_t delta) {
  std::lock_guard<std::mutex> lock(mu_);
  std::string req(1, (char)kAdd);
  put_str(req, key);
  req.append(reinterpret_cast<const char*>(&delta), 8);
  net::send_all(fd_, req.data(), req.size());
  int64_t v = 0;
  net::recv_all(fd_, &v, 8, timeout_s_);
  return v;
}

void StoreClient::erase_prefix(const std::string& prefix) {
  std::lock_guard<std::mutex> lock(mu_);
  std::string req(1, (char)kDel);
  put_str(req, prefix);
  net::send_all(fd_, req.data(), req.size());
  uint8_t ok = 0;
  net::recv_all(fd_, &ok, 1, timeout_s_);
}

std::string StoreClient::local_address() const { return net::local_address(fd_); }

// ---------------------------------------------------------------------------------- rendezvous
void Rendezvous::barrier(const std::string& tag) {
  if (world <= 1 || !client) return;
  const int64_t seq = barrier_seq++;
  const std::string base = "barrier/" + tag + "/" + std::to_string(seq);
  if (client->add(base + "/count", 1) == world) {
    // Last to arrive: every rank has entered round `seq`, so every rank has left round seq - 1
    // (its `done` was read) and those two keys can go.
    if (seq > 0) client->erase_prefix("barrier/" + tag + "/" + std::to_string(seq - 1) + "/");
    client->set(base + "/done", "1");
  }
  (void)client->get(base + "/done");
}

std::vector<std::string> Rendezvous::exchange(const std::string& tag, const std::string& mine) {
  if (world <= 1 || !client) return {mine};
  const int64_t seq = exchange_seq++;
  const std::string base = "xchg/" + tag + "/" + std::to_string(seq) + "/";
  client->set(base + std::to_string(rank), mine);
  std::vector<std::string> out;
  for (int r = 0; r < world; ++r) out.push_back(r == rank ? mine : client->get(base + std::to_string(r)));
  // Having read every rank's entry of this round, every rank has begun it and therefore finished
  // reading the round before: that round's entries have no reader left (rank 0 drops them).
  if (seq > 0 && rank == 0) client->erase_prefix("xchg/" + tag + "/" + std::to_string(seq - 1) + "/");
  return out;
}

std::shared_ptr<Rendezvous> rendezvous_from_env(int rank, int world, const std::string& host, int port, double timeout_s,
                                                const std::string& job_id) {
  auto rv = std::make_shared<Rendezvous>();
  rv->rank = rank;
  rv->world = world;
  rv->master_host = host;
  if (world <= 1) return rv;
  constexpr int kScan = 16;
  if (rank == 0) rv->server = std::make_unique<StoreServer>(host, port, kScan, job_id);
  // Rank 0 reaches its own server by the address it bound (loopback when it listens everywhere).
  const std::string own = rank == 0 && !rv->server->bound_host().empty() ? rv->server->bound_host() : "127.0.0.1";
  rv->client = std::make_unique<StoreClient>(rank == 0 ? own : host, rank == 0 ? rv->server->port() : port,
                                             rank == 0 ? 1 : kScan, job_id, timeout_s);
  return rv;
}

}  // namespace comm
}  // namespace pda
