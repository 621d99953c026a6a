// The opt-in MPI adapter (compose_amd/csrc/cedr_mpi_adapter.hpp) against a real MPI library:
// reference-style calling code -- make_parallel(MPI_Comm), `MPI_Comm c = p->comm()`, the typed
// wrappers of cedr/cedr_mpi.hpp:29-85 -- must compile unchanged and behave as MPI does.
// Host only (no GPU): the CDRs' device collectives through MpiComm are exercised on the GPU box.
#include <mpi.h>

#include <cstdio>
#include <vector>

#include "cedr_mpi_adapter.hpp"

#define REQUIRE(c) do { if ( ! (c)) { std::printf("rank %d FAIL %s:%d: %s\n", rank, __FILE__, __LINE__, #c); ++nerr; } } while (0)

static void my_max (void* in, void* inout, int* len, MPI_Datatype*) {
  for (int i = 0; i < *len; ++i)
    if (static_cast<double*>(in)[i] > static_cast<double*>(inout)[i]) static_cast<double*>(inout)[i] = static_cast<double*>(in)[i];
}

int main (int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int nerr = 0, rank = -1, size = 0;
  {
    using namespace cedr;
    const mpi::Parallel::Ptr p = mpi::make_parallel(MPI_COMM_WORLD);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    REQUIRE(p->rank() == rank && p->size() == size && p->root() == 0 && p->amroot() == (rank == 0));
    MPI_Comm c = p->comm();  // reference cedr_mpi.hpp:22
    int cmp = MPI_UNEQUAL;
    MPI_Comm_compare(c, MPI_COMM_WORLD, &cmp);
    REQUIRE(cmp == MPI_IDENT);
    REQUIRE(MPI_Barrier(p->comm()) == MPI_SUCCESS);
    // a Parallel over another transport has no MPI communicator
    MPI_Comm none = mpi::make_parallel()->comm();
    REQUIRE(none == MPI_COMM_NULL);

    double s = rank + 1.0, r = 0;
    REQUIRE(mpi::all_reduce(*p, &s, &r, 1, MPI_SUM) == MPI_SUCCESS && r == 0.5*size*(size + 1));
    long lv = rank, lr = -1;
    REQUIRE(mpi::reduce(*p, &lv, &lr, 1, MPI_MAX, 0) == MPI_SUCCESS && (rank != 0 || lr == size - 1));
    mpi::Op op(my_max, true);
    REQUIRE(mpi::all_reduce(*p, &s, &r, 1, op.get()) == MPI_SUCCESS && r == size);

    // ring: receive from the left with a Request, send to the right fire-and-forget
    std::vector<mpi::Request> reqs(1);
    int got = -1, mine = 100 + rank;
    REQUIRE(mpi::irecv(*p, &got, 1, (rank + size - 1) % size, 42, &reqs[0]) == MPI_SUCCESS);
    REQUIRE(mpi::isend(*p, &mine, 1, (rank + 1) % size, 42) == MPI_SUCCESS);
    REQUIRE(mpi::waitall(1, reqs.data()) == MPI_SUCCESS && got == 100 + (rank + size - 1) % size);
    int idx = -1;
    REQUIRE(mpi::irecv(*p, &got, 1, (rank + 1) % size, 43, &reqs[0]) == MPI_SUCCESS);
    REQUIRE(mpi::isend(*p, &mine, 1, (rank + size - 1) % size, 43) == MPI_SUCCESS);
    REQUIRE(mpi::waitany(1, reqs.data(), &idx) == MPI_SUCCESS && idx == 0 && got == 100 + (rank + 1) % size);

    std::vector<int> all(size, -1), counts(size), displs(size);
    REQUIRE(mpi::gather(*p, &mine, 1, all.data(), 1, 0) == MPI_SUCCESS);
    if (rank == 0) for (int i = 0; i < size; ++i) REQUIRE(all[i] == 100 + i);
    std::vector<double> part(rank + 1, rank), cat(size*(size + 1)/2, -1);
    for (int i = 0; i < size; ++i) { counts[i] = i + 1; displs[i] = i*(i + 1)/2; }
    REQUIRE(mpi::gatherv(*p, part.data(), rank + 1, cat.data(), counts.data(), displs.data(), 0) == MPI_SUCCESS);
    if (rank == 0) for (int i = 0; i < size; ++i) for (int j = 0; j <= i; ++j) REQUIRE(cat[displs[i] + j] == i);
    REQUIRE(mpi::all_ok(*p, true) && ! mpi::all_ok(*p, rank != size - 1));
  }
  int tot = 0;
  MPI_Allreduce(&nerr, &tot, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
  if (rank == 0) std::printf(tot ? "MPI_ADAPTER_FAIL %d\n" : "MPI_ADAPTER_OK %d ranks\n", tot ? tot : size);
  MPI_Finalize();
  return tot ? 1 : 0;
}
