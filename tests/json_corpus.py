"""Seeded corpus for the JSON differential tests: documents (valid and mutated) and numbers."""
import json
import random
import struct

HAND = ['{"a":1}', '[]', '{}', '[1,2.0,-0.0,1e5,1E-7,123456789012345678,-9223372036854775808,18446744073709551615,1.5e300,0.1]',
        '"\\u00e9\\n\\t\\"\\\\\\/\\b\\f\\r"', '{"b":{"a":[{}, [], [[]], {"x":null}]}, "a":true, "c":false}', '  [ 1 , 2 ]  ',
        '{"k":"\\ud83d\\ude00 é ü"}', '[1e400]', '[0.20000000298023224, 1.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308]',
        '{"a":1,"a":2}', '[01]', '[1.]', '[.5]', '{"a":}', '[1,]', 'nul', '"abc', '[-]', '[1e]', '{"SceneName":"", "SceneObjects":[]}', '3', '-0',
        '"\\u0000\\u001f"', '[1.0e+2, 1e-2, 100, 1E2]', '\ufeff[1]', '[1] x', '"\x01"', '["\t"]',
        '[18446744073709551616, -9223372036854775809, 1e22, 1e21, 123456789.0, 0.000001, 1e-5]', '"\\ud800"', '"\\udc00x"', '["\x7f", " "]',
        '[true,false,null]', '{"":0}', '{"a\\nb":1}', '[1e-931]', '[-1e-400, 4.9e-324, 2e-324]', '"\xed\xa0\x80"', '"\xc0\xaf"', '"\xf4\x90\x80\x80"', '"\xe2\x82"']


def _rfloat(rnd):
    k = rnd.random()
    if k < 0.5:
        return struct.unpack('<f', struct.pack('<I', rnd.getrandbits(32)))[0]
    if k < 0.8:
        return struct.unpack('<d', struct.pack('<Q', rnd.getrandbits(64)))[0]
    return rnd.choice([0.0, -0.0, 1.0, 0.1, 1e21, 1e22, 1e-5, 1e-7, 123456.789, 5e-324, 1.7976931348623157e308])


def _rstr(rnd):
    return ''.join(rnd.choice(['a', 'Z', '"', '\\', '/', '\n', '\t', '\x01', '\x1f', 'é', 'ü', '€', '😀', ' ', '0', '\x7f', '\u2028'])
                   for _ in range(rnd.randint(0, 8)))


def _rval(rnd, d=0):
    k = rnd.random()
    if d > 3 or k < 0.35:
        c = rnd.random()
        if c < 0.4:
            f = _rfloat(rnd)
            return f if f == f and abs(f) != float('inf') else 0.5
        if c < 0.6:
            return rnd.choice([0, 1, -1, 2**31, -2**63, 2**64 - 1, 2**63, rnd.getrandbits(40)])
        if c < 0.8:
            return _rstr(rnd)
        return rnd.choice([True, False, None])
    if k < 0.65:
        return [_rval(rnd, d + 1) for _ in range(rnd.randint(0, 4))]
    return {_rstr(rnd): _rval(rnd, d + 1) for _ in range(rnd.randint(0, 4))}


def documents(n, seed=7):
    """n single-line documents as bytes: generated JSON, about a third of them byte-mutated."""
    rnd = random.Random(seed)
    out = [h.encode('utf-8') if isinstance(h, str) and all(ord(c) < 0x80 or ord(c) > 0xff for c in h) else
           h.encode('latin-1') if all(ord(c) <= 0xff for c in h) else h.encode('utf-8') for h in HAND]
    while len(out) < n:
        t = json.dumps(_rval(rnd), ensure_ascii=rnd.random() < 0.5).encode('utf-8')
        if rnd.random() < 0.35:
            b = bytearray(t)
            for _ in range(rnd.randint(1, 3)):
                if not b:
                    break
                j = rnd.randrange(len(b))
                op = rnd.random()
                if op < 0.4:
                    b[j] = rnd.choice(b' {}[],:"\\0123456789eE+-.tfnu\xff\xc3')
                elif op < 0.7:
                    del b[j]
                else:
                    b.insert(j, rnd.choice(b' {}[],:"\\0123456789eE+-.'))
            t = bytes(b)
        if b'\n' in t or b'\r' in t or b'\x00' in t:
            continue
        out.append(t)
    return [d for d in out if b'\n' not in d and b'\r' not in d and b'\x00' not in d][:n]


def numbers(n, seed=11):
    """n finite doubles: random binary32 values (what the reference stores), random binary64 and edge cases."""
    rnd = random.Random(seed)
    out = [0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 1e21, 1e22, 1e-5, 1e-7, 123456.789, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
           9007199254740992.0, 9007199254740993.0, 1e15, 1e16, 1e17, 0.001, 0.0001, 0.00001, 299792458.0, 0.20000000298023224]
    while len(out) < n:
        if rnd.random() < 0.7:
            v = struct.unpack('<f', struct.pack('<I', rnd.getrandbits(32)))[0]
        else:
            v = struct.unpack('<d', struct.pack('<Q', rnd.getrandbits(64)))[0]
        if v == v and abs(v) != float('inf'):
            out.append(float(v))
    return out
