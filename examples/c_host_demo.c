/*
 * c_host_demo.c — the C ABI used from plain C: no Python, no torch.  This is what the JNI shim
 * (integration/jni/surge_replay_jni.c) does on behalf of the JVM.
 *
 *   gcc -std=c99 -Iinclude examples/c_host_demo.c -Lsurge_amd -lsurge_replay -Wl,-rpath,$PWD/surge_amd -o /tmp/c_host_demo
 *
 * Replays the reference's own known answers (PersistentActorSpec.scala:134-168, 466-493;
 * BankAccountCommandEngineSpec.scala:44-68; a throwing event, PersistentActorSpec.scala:431-464) and
 * prints PASS / FAIL.  Exit code 0 = all good, 2 = no GPU (the library has no CPU fallback).
 */
#include <stdio.h>
#include <string.h>
#include "surge_replay.h"

static surge_event16 ev_int(int type, int seq, int arg) {
  surge_event16 e;
  memset(&e, 0, sizeof(e));
  e.type = type; e.seq = seq; e.p.i.arg = arg;
  return e;
}
static surge_event16 ev_f64(int type, int seq, double v) {
  surge_event16 e;
  memset(&e, 0, sizeof(e));
  e.type = type; e.seq = seq; e.p.value = v;
  return e;
}

int main(void) {
  surge_replay_schema sc;
  surge_replay_handle* h = NULL;
  surge_state64 init[4], out[4];
  surge_event16 ev[16];
  int64_t seg_off[5];
  uint8_t present[4];
  int n = 0, rc, fails = 0;

  surge_replay_default_schema(&sc);
  rc = surge_replay_create(&sc, 0, &h);
  if (rc == SURGE_E_DEVICE) { printf("no GPU: %s\n", surge_replay_last_error(NULL)); return 2; }
  if (rc != SURGE_OK) { printf("create failed: %s\n", surge_replay_last_error(NULL)); return 1; }

  memset(init, 0, sizeof(init));
  /* aggregate 0: State(id,3,3) + two Increments -> (5,5) */
  init[0].count = 3; init[0].version = 3; init[0].min_arg = 0x7fffffff; init[0].max_arg = (int32_t)0x80000000;
  init[0].flags = SURGE_STATE_PRESENT;
  seg_off[0] = n;
  ev[n++] = ev_int(SURGE_EVT_INC, 4, 1);
  ev[n++] = ev_int(SURGE_EVT_INC, 5, 1);
  /* aggregate 1: BankAccount created with 1000.0 then balance set to 1100.0 */
  seg_off[1] = n;
  ev[n++] = ev_f64(SURGE_EVT_CREATE, 0, 1000.0);
  ev[n++] = ev_f64(SURGE_EVT_SET_BALANCE, 0, 1000.0 + 100.0);
  /* aggregate 2: an update before any create is dropped: stays None */
  seg_off[2] = n;
  ev[n++] = ev_f64(SURGE_EVT_SET_BALANCE, 0, 5.0);
  /* aggregate 3: increment, throwing event, increment -> frozen at (1,1), poisoned */
  seg_off[3] = n;
  ev[n++] = ev_int(SURGE_EVT_INC, 1, 1);
  ev[n++] = ev_int(SURGE_EVT_THROW, 2, 0);
  ev[n++] = ev_int(SURGE_EVT_INC, 3, 1);
  seg_off[4] = n;

  rc = surge_replay_load_csr(h, seg_off, 4, ev, n, init);
  if (rc == SURGE_OK) rc = surge_replay_fold(h, SURGE_ALGO_AUTO);
  if (rc == SURGE_OK) rc = surge_replay_snapshot(h, out, present);
  if (rc != SURGE_OK) { printf("replay failed: %s\n", surge_replay_last_error(h)); return 1; }

#define CHECK(cond, what) do { if (!(cond)) { ++fails; printf("FAIL %s\n", what); } else printf("PASS %s\n", what); } while (0)
  CHECK(out[0].count == 5 && out[0].version == 5 && present[0], "two increments on (3,3) -> (5,5)");
  CHECK(out[1].balance == 1100.0 && present[1], "create 1000.0 then set 1100.0");
  CHECK(!present[2] && out[2].flags == 0, "update before create stays None");
  CHECK(out[3].count == 1 && out[3].version == 1 && (out[3].flags & SURGE_STATE_POISONED), "throwing event freezes the state before it");
  {
    surge_state64 one; uint8_t p = 0;
    rc = surge_replay_get(h, 1, &one, &p);
    CHECK(rc == SURGE_OK && p && one.balance == 1100.0, "point read after snapshot");
    CHECK(surge_replay_get(h, 9, &one, &p) == SURGE_E_RANGE, "out-of-range read is an error");
  }
  {
    /* one host, two handles (one per GPU on a multi-GPU node; both on device 0 here): the snapshot exchange as peer
     * copies, no RCCL, no rendezvous — every handle ends up with every handle's shard */
    surge_replay_handle* h2 = NULL;
    surge_replay_handle* group[2];
    surge_event16 ev2[2];
    int64_t off2[3] = {0, 1, 2};
    surge_state64 got[4];
    ev2[0] = ev_int(SURGE_EVT_INC, 1, 40);
    ev2[1] = ev_int(SURGE_EVT_INC, 1, 41);
    rc = surge_replay_create(&sc, 0, &h2);
    if (rc == SURGE_OK) rc = surge_replay_load_csr(h2, off2, 2, ev2, 2, NULL);
    if (rc == SURGE_OK) rc = surge_replay_fold(h2, SURGE_ALGO_AUTO);
    group[0] = h; group[1] = h2;
    if (rc == SURGE_OK) rc = surge_replay_allgather(group, 2, NULL, NULL, 0, 0);
    if (rc == SURGE_OK) rc = surge_replay_gathered_read(h, 0, 1, 0, 4, got);   /* handle 0 reads handle 1's shard */
    CHECK(rc == SURGE_OK && got[0].count == 40 && got[1].count == 41 && got[2].flags == 0 && got[3].flags == 0,
          "in-process exchange: the other handle's shard, padded with None up to the largest shard");
    if (rc == SURGE_OK) rc = surge_replay_gathered_read(h2, 0, 0, 0, 4, got);  /* and the other way round */
    CHECK(rc == SURGE_OK && got[0].count == 5 && got[1].balance == 1100.0, "in-process exchange: both directions");
    if (h2) surge_replay_destroy(h2);
  }
  surge_replay_destroy(h);
  return fails ? 1 : 0;
}
