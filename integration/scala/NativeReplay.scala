// SOURCE-ONLY (no JVM/sbt in the build image).  JNI binding of include/surge_replay.h; the native
// side is integration/jni/surge_replay_jni.c.
package surge.replay.gpu

import java.nio.ByteBuffer

object NativeReplay {
  System.loadLibrary("surge_replay_jni") // links libsurge_replay.so (the HIP engine)

  @native def create(schema: ByteBuffer /* null = built-in algebra */, device: Int): Long
  @native def destroy(handle: Long): Unit
  @native def loadCsr(handle: Long, segOff: ByteBuffer, nAgg: Long, events: ByteBuffer, nEvents: Long, initState: ByteBuffer): Int
  @native def fold(handle: Long, algo: Int): Int
  @native def appendFold(handle: Long, groupAgg: ByteBuffer, groupOff: ByteBuffer, nGroups: Long, events: ByteBuffer, nEvents: Long): Int
  @native def snapshot(handle: Long, states: ByteBuffer, present: ByteBuffer): Int
  @native def get(handle: Long, aggIdx: Long, state64: ByteBuffer): Int
  @native def partitionHash(utf16: ByteBuffer, strOff: ByteBuffer, n: Long, nPartitions: Int, partOut: ByteBuffer): Int
}
