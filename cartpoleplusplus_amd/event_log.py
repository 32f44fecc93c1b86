"""Episode event log with the on-disk format of the reference's event_log.py / event.proto:
length-prefixed (`struct '=l'`) proto2 `Episode{ Event{ action[], State{cart_pose[], pole_pose[],
Render{height,width,png_bytes}[]}[], reward } }` frames, pixel renders PNG encoded.  Feeds
`ReplayMemory.reset_from_event_log` (replay_memory.py:40-61) and `--event-log-in` / `--dont-do-rollouts`.

Reference: /root/reference/event.proto:1-35, event_log.py:10-111.  The message classes are built at import time
from a programmatic descriptor (the reference generates event_pb2 with protoc, which is not a dependency here);
PNG bytes are read / written by a small zlib-based codec for the 8-bit RGB(A), non-interlaced files that
matplotlib's imsave produces (event_log.py:10-14), so neither matplotlib nor PIL is required.
"""
import gzip
import os
import struct
import sys
import zlib

import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory


def _build_messages():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "cartpolepp_event.proto", "cp", "proto2"
    F = descriptor_pb2.FieldDescriptorProto

    def msg(name, fields):
        m = fd.message_type.add()
        m.name = name
        for fname, num, ftype, label, tname in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, num, ftype, label
            if tname:
                f.type_name = ".cp." + tname

    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("Render", [("height", 1, F.TYPE_INT32, OPT, None), ("width", 2, F.TYPE_INT32, OPT, None),
                   ("png_bytes", 3, F.TYPE_BYTES, OPT, None)])
    msg("State", [("cart_pose", 1, F.TYPE_FLOAT, REP, None), ("pole_pose", 2, F.TYPE_FLOAT, REP, None),
                  ("render", 3, F.TYPE_MESSAGE, REP, "Render")])
    msg("Event", [("action", 1, F.TYPE_FLOAT, REP, None), ("state", 2, F.TYPE_MESSAGE, REP, "State"),
                  ("reward", 3, F.TYPE_FLOAT, OPT, None)])
    msg("Episode", [("event", 1, F.TYPE_MESSAGE, REP, "Event")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("cp." + n))
    return get("Render"), get("State"), get("Event"), get("Episode")


Render, State, Event, Episode = _build_messages()

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def rgb_to_png(rgb):
    """(H, W, 3) floats in [0, 1] -> RGBA 8-bit PNG bytes (what plt.imsave writes, event_log.py:10-14)."""
    rgb = np.asarray(rgb, dtype=np.float64)
    h, w = rgb.shape[:2]
    rgba = np.empty((h, w, 4), np.uint8)
    rgba[:, :, :3] = np.clip(np.rint(rgb * 255.0), 0, 255).astype(np.uint8)
    rgba[:, :, 3] = 255
    raw = b"".join(b"\x00" + rgba[y].tobytes() for y in range(h))           # filter type 0 per scanline

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    return (_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def png_to_rgb(png_bytes):
    """PNG bytes -> (H, W, 3) float32 in [0, 1] (k/255), alpha dropped (event_log.py:16-20)."""
    assert png_bytes[:8] == _PNG_SIG, "not a PNG"
    pos, idat, ihdr = 8, [], None
    while pos < len(png_bytes):
        n, tag = struct.unpack(">I4s", png_bytes[pos:pos + 8])
        data = png_bytes[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"IDAT":
            idat.append(data)
        elif tag == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _comp, _filt, interlace = ihdr
    assert depth == 8 and ctype in (2, 6) and interlace == 0, "only 8-bit RGB/RGBA non-interlaced PNGs"
    bpp = 4 if ctype == 6 else 3
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * bpp)
    out = np.zeros((h, w * bpp), np.uint8)
    prev = np.zeros(w * bpp, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:                                   # Up
            cur = (line + prev) & 0xff
        else:                                           # Sub / Average / Paeth need the running left neighbour
            cur = np.zeros_like(line)
            for x in range(w * bpp):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + pred) & 0xff
        out[y] = cur
        prev = cur
    return (out.reshape(h, w, bpp)[:, :, :3].astype(np.float32) / np.float32(255.0))


def read_state_from_event(event):
    """unpack state from event (event_log.py:21-39): pixels -> (H, W, 3, cameras, repeats); poses -> (repeats, 2, 7)."""
    if len(event.state[0].render) > 0:
        num_repeats, num_cameras = len(event.state), len(event.state[0].render)
        eg = event.state[0].render[0]
        state = np.empty((eg.height, eg.width, 3, num_cameras, num_repeats))
        for r_idx in range(num_repeats):
            for c_idx in range(num_cameras):
                state[:, :, :, c_idx, r_idx] = png_to_rgb(event.state[r_idx].render[c_idx].png_bytes)
    else:
        state = np.empty((len(event.state), 2, 7))
        for i, s in enumerate(event.state):
            state[i][0] = s.cart_pose
            state[i][1] = s.pole_pose
    return state


class EventLog(object):
    """writer (event_log.py:41-93): one Episode per env.reset()."""

    def __init__(self, path, use_raw_pixels):
        self.log_file = open(path, "ab")
        self.episode_entry = None
        self.use_raw_pixels = use_raw_pixels

    def reset(self):
        if self.episode_entry is not None:
            buff = self.episode_entry.SerializeToString()
            if len(buff) > 0:
                self.log_file.write(struct.pack('=l', len(buff)))
                self.log_file.write(buff)
                self.log_file.flush()
        self.episode_entry = Episode()

    def add_state_to_event(self, state, event):
        state = np.asarray(state)
        if self.use_raw_pixels:
            for r_idx in range(state.shape[4]):
                s = event.state.add()
                for c_idx in range(state.shape[3]):
                    render = s.render.add()
                    render.width, render.height = state.shape[1], state.shape[0]
                    render.png_bytes = rgb_to_png(state[:, :, :, c_idx, r_idx])
        else:
            for r in range(state.shape[0]):
                s = event.state.add()
                s.cart_pose.extend(float(x) for x in state[r][0])
                s.pole_pose.extend(float(x) for x in state[r][1])

    def add(self, state, action, reward):
        event = self.episode_entry.event.add()
        self.add_state_to_event(state, event)
        if isinstance(action, int):
            event.action.append(action)
        else:
            action = np.asarray(action)
            assert action.shape[0] == 1                 # never log batch operations
            event.action.extend(float(x) for x in action[0])
        event.reward = reward

    def add_just_state(self, state):
        self.add_state_to_event(state, self.episode_entry.event.add())

    def close(self):
        self.reset()
        self.log_file.close()


class EventLogReader(object):
    def __init__(self, path):
        self.log_file = gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")

    def entries(self):
        while True:
            head = self.log_file.read(4)
            if len(head) == 0:
                return
            episode = Episode()
            episode.ParseFromString(self.log_file.read(struct.unpack('=l', head)[0]))
            yield episode


def make_dir(d):
    if not os.path.exists(d):
        os.makedirs(d)


def _draw_line(img, x0, y0, x1, y1):
    n = int(max(abs(x1 - x0), abs(y1 - y0), 1))
    for i in range(n + 1):
        x, y = int(round(x0 + (x1 - x0) * i / n)), int(round(y0 + (y1 - y0) * i / n))
        if 0 <= y < img.shape[0] and 0 <= x < img.shape[1]:
            img[y, x, :3] = 0.0


def main(argv=None):
    """the log inspector at the bottom of the reference's event_log.py (:118-190): `--echo` prints the episodes, `--img-output-dir DIR`
    writes every render to DIR/ep_NNNNN/cK/eNNNNN_rK.png (at the render's own size: the reference upscales to 200 x 200 with PIL),
    `--img-debug-overlay` draws the action as a line inside a box at (40, 40) (the reference also prints the episode / event numbers
    as text), `--episodes 0,3` restricts both."""
    import argparse
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('--log-file', type=str, default=None)
    parser.add_argument('--echo', action='store_true', help="write event to stdout")
    parser.add_argument('--episodes', type=str, default=None,
                        help="if set only process these specific episodes (comma separated list)")
    parser.add_argument('--img-output-dir', type=str, default=None,
                        help="if set output all renders to this DIR/e_NUM/s_NUM.png")
    parser.add_argument('--img-debug-overlay', action='store_true', help="if set overlay image with debug info")
    o = parser.parse_args(argv)
    whitelist = None if o.episodes is None else set(map(int, o.episodes.split(",")))
    if o.img_output_dir is not None:
        make_dir(o.img_output_dir)
    n_episodes = n_events = 0
    for episode_id, episode in enumerate(EventLogReader(o.log_file).entries()):
        if whitelist is not None and episode_id not in whitelist:
            continue
        if o.echo:
            print("-----", episode_id)
            print(episode)
        n_episodes += 1
        n_events += len(episode.event)
        if o.img_output_dir is not None:
            d = "%s/ep_%05d" % (o.img_output_dir, episode_id)
            for sub in ("", "/c0", "/c1"):
                make_dir(d + sub)
            for event_id, event in enumerate(episode.event):
                for state_id, state in enumerate(event.state):
                    for camera_id, render in enumerate(state.render):
                        assert camera_id in [0, 1], "two cameras at most"
                        png = render.png_bytes
                        if o.img_debug_overlay:
                            img = np.array(png_to_rgb(png), np.float64)
                            bx, by, bw = 40, 40, 10
                            for (x0, y0, x1, y1) in ((bx - bw, by - bw, bx + bw, by - bw), (bx + bw, by - bw, bx + bw, by + bw),
                                                     (bx + bw, by + bw, bx - bw, by + bw), (bx - bw, by + bw, bx - bw, by - bw)):
                                _draw_line(img, x0, y0, x1, y1)
                            if len(event.action) >= 2:
                                _draw_line(img, bx, by, bx + event.action[0] * bw, by + event.action[1] * bw)
                            png = rgb_to_png(img[:, :, :3])
                        with open("%s/c%d/e%05d_r%d.png" % (d, camera_id, event_id, state_id), "wb") as f:
                            f.write(png)
    print("read", n_episodes, "episodes for a total of", n_events, "events", file=sys.stderr)
    return n_episodes, n_events


if __name__ == "__main__":
    main()
