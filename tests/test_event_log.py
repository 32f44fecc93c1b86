"""CPU tests of the event-log format (event.proto / event_log.py of the reference): frame layout, proto
round trip, PNG codec against PIL (an independent decoder/encoder), and episode structure."""
import io
import struct

import numpy as np
import pytest

from cartpoleplusplus_amd import event_log as E


def frames(rng, shape, n):
    return [(rng.integers(0, 256, shape).astype(np.float16) / np.float16(255)).astype(np.float32) for _ in range(n)]


def test_png_codec_matches_pil_both_ways():
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (7, 9, 3)).astype(np.uint8)
    ours = E.rgb_to_png(img.astype(np.float32) / 255)
    back = np.asarray(PIL.open(io.BytesIO(ours)).convert("RGB"))
    assert np.array_equal(back, img)
    # PIL-written PNGs use adaptive filters (Sub/Up/Average/Paeth): exercises every branch of the decoder
    for mode in ("RGB", "RGBA"):
        smooth = (np.add.outer(np.arange(16), np.arange(20))[:, :, None] * np.array([3, 5, 7])).astype(np.uint8)
        buf = io.BytesIO()
        im = PIL.fromarray(smooth, "RGB").convert(mode)
        im.save(buf, format="PNG")
        got = E.png_to_rgb(buf.getvalue())
        assert got.dtype == np.float32 and np.array_equal(got, smooth.astype(np.float32) / np.float32(255))


def test_pixel_episode_round_trip_and_frame_layout(tmp_path):
    rng = np.random.default_rng(1)
    shape = (12, 10, 3, 2, 3)
    path = str(tmp_path / "events")
    log = E.EventLog(path, use_raw_pixels=True)
    episodes = []
    for ep in range(3):
        log.reset()
        fr = frames(rng, shape, 4)
        log.add_just_state(fr[0])
        acts = [rng.uniform(-1, 1, (1, 2)).astype(np.float32) for _ in range(3)]
        for k in range(3):
            log.add(fr[k + 1], acts[k], float(k + 1))
        episodes.append((fr, acts))
    log.close()
    raw = open(path, "rb").read()
    n0 = struct.unpack("=l", raw[:4])[0]                    # event_log.py:52-55: 4-byte native length prefix
    assert 0 < n0 < len(raw)
    got = list(E.EventLogReader(path).entries())
    assert len(got) == 3
    for epi, (fr, acts) in zip(got, episodes):
        assert len(epi.event) == 4
        assert len(epi.event[0].action) == 0 and not epi.event[0].HasField("reward")
        assert len(epi.event[1].state) == 3 and len(epi.event[1].state[0].render) == 2
        for k, ev in enumerate(epi.event):
            st = E.read_state_from_event(ev)
            # the PNG keeps the 8-bit level k; the replay memory's f16 cast maps k/255 back onto the env's f16(k/255)
            assert st.shape == shape and np.array_equal(st.astype(np.float16), fr[k].astype(np.float16))
            if k:
                assert np.allclose(np.asarray(ev.action), acts[k - 1][0]) and ev.reward == float(k)


def test_lowdim_episode_round_trip(tmp_path):
    rng = np.random.default_rng(2)
    path = str(tmp_path / "events_lowdim")
    log = E.EventLog(path, use_raw_pixels=False)
    log.reset()
    states = [rng.standard_normal((2, 2, 7)).astype(np.float32) for _ in range(3)]
    log.add_just_state(states[0])
    log.add(states[1], np.array([[0.25, -0.5]], np.float32), 1.0)
    log.add(states[2], np.array([[0.75, 0.125]], np.float32), 1.0)
    log.close()
    (epi,) = list(E.EventLogReader(path).entries())
    for k, ev in enumerate(epi.event):
        assert np.array_equal(E.read_state_from_event(ev).astype(np.float32), states[k])
    assert list(epi.event[2].action) == [0.75, 0.125]


def test_log_inspector_cli(tmp_path, capsys):
    """the reader script at the bottom of event_log.py (:118-190): --echo, --episodes, --img-output-dir, --img-debug-overlay"""
    import os
    from cartpoleplusplus_amd import event_log as E
    path = str(tmp_path / "events")
    rng = np.random.default_rng(0)
    log = E.EventLog(path, use_raw_pixels=True)
    shape = (50, 50, 3, 2, 2)
    for _ep in range(3):
        log.reset()
        log.add_just_state(rng.integers(0, 256, shape).astype(np.float32) / 255)
        for _ in range(4):
            log.add(rng.integers(0, 256, shape).astype(np.float32) / 255, rng.uniform(-1, 1, (1, 2)).astype(np.float32), 1.0)
    log.close()
    out_dir = str(tmp_path / "imgs")
    n_ep, n_ev = E.main(["--log-file", path, "--echo", "--episodes", "0,2", "--img-output-dir", out_dir, "--img-debug-overlay"])
    assert (n_ep, n_ev) == (2, 10)
    text = capsys.readouterr().out
    assert "----- 0" in text and "----- 2" in text and "----- 1" not in text
    assert sorted(os.listdir(out_dir)) == ["ep_00000", "ep_00002"]
    files = sorted(os.listdir(out_dir + "/ep_00002/c1"))
    assert len(files) == 5 * 2 and files[0] == "e00000_r0.png"
    img = E.png_to_rgb(open(out_dir + "/ep_00002/c1/e00003_r1.png", "rb").read())
    assert img.shape[:2] == (50, 50) and (np.asarray(img)[30, 30:51, :3] == 0).all()      # the overlay's box edge
