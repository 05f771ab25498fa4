// tools/MurmurPin.scala — generates the Scala-runtime pin for the shard map (SURVEY §8c: "parity unpinned").
//
//   scala tools/MurmurPin.scala > tests/golden/murmur_pin.tsv        (scala-library 2.13.x, like build.sbt:6)
//
// One line per probe string:  <hex of the UTF-16 code units> TAB stringHash TAB partitionForKey(s, 64)
// with partitionForKey exactly as modules/common/src/main/scala/surge/kafka/KafkaPartitioner.scala:8.
// tests/test_oracle_kat.py::test_scala_runtime_pin_of_the_shard_map compares the oracle AND the product entry
// points with this file when it exists; the test regenerates it when `scala` is on PATH and is skipped otherwise.
// No JVM exists in the build image, so the file is absent until a maintainer runs this once.
import scala.util.hashing.MurmurHash3

object MurmurPin {
  def partitionForKey(partitionByString: String, numberOfPartitions: Int): Int =
    math.abs(MurmurHash3.stringHash(partitionByString) % numberOfPartitions)

  def main(args: Array[String]): Unit = {
    val probes = Seq("", "a", "ab", "abc", "abcd", "acct-00000000", "acct-00000042", "acct-09999999", "stateKey1",
      "stateKey1:17", "aggregate:with:colons", "::", "CounterAggregate", "ünï-✓", "😀x") ++
      (0 until 64).map(i => f"acct-$i%08d")
    probes.foreach { s =>
      val hex = s.map(c => f"${c.toInt}%04x").mkString
      println(s"$hex\t${MurmurHash3.stringHash(s)}\t${partitionForKey(s, 64)}")
    }
  }
}
