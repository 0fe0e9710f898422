"""C4 crowd frames with the two workgroup -> XCD orders of the instanced skin kernel (tuning key inst_order), in the three pose modes."""
import json, subprocess, sys
for mode in ([], ["--device-fk"], ["--device-fk", "--device-sampling"]):
    for o in (0, 1):
        out = subprocess.run([sys.executable, "bench.py", "--config", "c4", "--no-cpu-baseline", "--no-sampled-loop", "--tune", "inst_order=%d" % o] + mode,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
        c = d["config"]
        print("%-18s inst_order=%d  step %.2f us  kernel %.2f us  frame(events) %.2f us  front kernels %.2f us  with pose upload %s  sampled loop %s" % (
            " ".join(mode) or "world matrices", o, d["ms_per_step"] * 1e3, d["roofline"]["kernel_ms"] * 1e3, c["frame_ms_events"] * 1e3, c["prep_kernel_ms"] * 1e3,
            c["frame_ms_with_pose_upload"] and round(c["frame_ms_with_pose_upload"] * 1e3, 2), c["frame_ms_device_sampled_pose"] and round(c["frame_ms_device_sampled_pose"] * 1e3, 2)), flush=True)
