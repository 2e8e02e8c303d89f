"""Second, independent CPU restatement of the reference's Raft hot path (test infrastructure:
only tests/ imports it).  `raft` follows the Rust line by line; `engine` wraps G of those
groups behind the interface the parity helpers drive (the row vocabulary of
include/josefine_gpu.h)."""
