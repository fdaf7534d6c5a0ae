// Rendezvous key-value store of the framework's own communication layer.
//
// One rank (rank 0) hosts a StoreServer thread; every rank, rank 0 included, talks to it
// through a StoreClient over TCP. It carries what a communicator needs before it exists:
// the RCCL unique id, the listening addresses of the CPU socket mesh, barriers and small
// agreement checks. Blocking `get` (the server parks the request until the key is set), an
// atomic `add` and a prefix `erase` are the only primitives beyond `set`. Barriers and
// exchanges erase the keys of the round before the one that just completed, so a long run
// (one exchange per collective under PDA_COMM_DEBUG=1) keeps rank 0's map at a few entries.
//
// Address: MASTER_ADDR, port MASTER_PORT + 1 by default (MASTER_PORT itself belongs to the
// launcher's own store). The server listens on the address MASTER_ADDR resolves to (every
// interface only when that address cannot be bound locally). When the port is taken, rank 0
// walks upwards and the clients find the right server through the handshake, which carries a
// job identifier.
//
// Role in the reference: the TCPStore rendezvous hidden inside
// dist.init_process_group(backend='nccl') (assignments/assignment1/train_ddp.py:23-24).
#pragma once

#include <atomic>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace pda {
namespace comm {

// Small socket helpers shared by the store and the CPU socket communicator.
namespace net {
// Listens on `host`'s address (empty, unresolvable or not local: every interface). Returns fd or -1
// (port in use); `*bound_host` receives the numeric address that was bound, "" for every interface.
int listen_on(const std::string& host, int port, int backlog, std::string* bound_host = nullptr);
int bound_port(int fd);
int connect_to(const std::string& host, int port, double timeout_s);  // retries until the deadline; -1 on failure
void send_all(int fd, const void* data, size_t n);
void recv_all(int fd, void* data, size_t n, double timeout_s);  // throws on timeout / peer close
void set_nodelay(int fd);
std::string local_address(int fd);  // the local IP of a connected socket
}  // namespace net

class StoreServer {
 public:
  // Binds `base_port`, or the first free port of the next `scan` ones.
  StoreServer(const std::string& host, int base_port, int scan, std::string job_id);
  ~StoreServer();
  int port() const { return port_; }
  const std::string& bound_host() const { return bound_host_; }  // "" = every interface
  size_t num_keys();  // entries currently held (tests: the map stays bounded)

 private:
  struct Client {
    int fd = -1;
    bool greeted = false;
    std::vector<uint8_t> in;
  };
  void loop();
  bool consume(Client& c);  // parses as many complete requests as `c.in` holds; false = drop client
  void answer_waiters(const std::string& key);

  std::string job_id_, bound_host_;
  std::mutex data_mu_;  // num_keys() reads the map from another thread
  int listen_fd_ = -1, wake_fd_[2] = {-1, -1}, port_ = 0;
  std::thread thread_;
  std::atomic<bool> stop_{false};
  std::atomic<int> open_clients_{0};  // connections currently open (rank 0's own included)
  std::map<std::string, std::string> data_;
  std::map<std::string, std::vector<int>> waiters_;
  std::map<int, Client> clients_;
};

class StoreClient {
 public:
  // Tries base_port .. base_port+scan-1 until a server answers the handshake for `job_id`.
  StoreClient(const std::string& host, int base_port, int scan, const std::string& job_id, double timeout_s);
  ~StoreClient();
  void set(const std::string& key, const std::string& value);
  std::string get(const std::string& key);  // blocks until the key exists (or the timeout passes)
  int64_t add(const std::string& key, int64_t delta);
  void erase_prefix(const std::string& prefix);  // drops every key that starts with `prefix`
  std::string local_address() const;
  double timeout_s() const { return timeout_s_; }

 private:
  int fd_ = -1;
  double timeout_s_;
  std::mutex mu_;
};

// Both ends of the rendezvous for one process.
struct Rendezvous {
  // Declaration order matters: members are destroyed in reverse, so rank 0's own client closes its
  // connection before the server starts waiting for the remaining (peer) connections to close.
  std::unique_ptr<StoreServer> server;  // rank 0 only
  std::unique_ptr<StoreClient> client;  // nullptr when world == 1
  int rank = 0, world = 1;
  std::string master_host;  // the address peers reach rank 0 by
  std::atomic<int64_t> barrier_seq{0}, exchange_seq{0};
  void barrier(const std::string& tag);
  // Every rank contributes a byte string and receives all of them in rank order.
  std::vector<std::string> exchange(const std::string& tag, const std::string& mine);
};

std::shared_ptr<Rendezvous> rendezvous_from_env(int rank, int world, const std::string& host, int port, double timeout_s,
                                                const std::string& job_id);

}  // namespace comm
}  // namespace pda
