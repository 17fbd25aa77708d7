"""HOCON-subset parser for the reference's configs/*.conf (pyhocon is not installed here): nested `name { ... }` blocks
(brace on the same or the next line), `key = value` / `key : value`, lists, unquoted / quoted strings, numbers, booleans,
`#` and `//` comments.  The returned Config mimics the pyhocon ConfigTree accessors the reference calls
(trainer/generate_ir_texture.py:36-65, trainer/train_material.py:36-39,97-128, models/mat_nvdiffrast.py:41-47)."""
import re


class ConfigMissing(KeyError):
    pass


_NO = object()


class Config(dict):
    def _get(self, key, default=_NO):
        cur = self
        for part in key.split("."):
            if isinstance(cur, dict) and part in cur:
                cur = cur[part]
            else:
                if default is _NO:
                    raise ConfigMissing("No configuration setting found for key %s" % key)
                return default
        return cur

    def get_string(self, key, default=_NO):
        v = self._get(key, default)
        return v if v is None else str(v)

    def get_int(self, key, default=_NO):
        v = self._get(key, default)
        return v if v is None else int(v)

    def get_float(self, key, default=_NO):
        v = self._get(key, default)
        return v if v is None else float(v)

    def get_bool(self, key, default=_NO):
        v = self._get(key, default)
        if isinstance(v, str):
            return v.lower() in ("true", "yes", "on")
        return bool(v)

    def get_list(self, key, default=_NO):
        v = self._get(key, default)
        return v if v is None else list(v)

    def get_config(self, key, default=_NO):
        v = self._get(key, default)
        if v is not None and not isinstance(v, dict):
            raise TypeError("%s is not a config block" % key)
        return v

    def get(self, key, default=None):
        return self._get(key, default)


_TOKEN = re.compile(r'''\s*(?:(?P<brace>[{}\[\],=:])|"(?P<q>(?:[^"\\]|\\.)*)"|(?P<w>[^\s{}\[\],=:"#]+))''')


def _scalar(w):
    lw = w.lower()
    if lw in ("true", "yes", "on"):
        return True
    if lw in ("false", "no", "off"):
        return False
    if lw == "null":
        return None
    try:
        return int(w)
    except ValueError:
        pass
    try:
        return float(w)
    except ValueError:
        return w


def _tokens(text):
    out = []
    for line in text.splitlines():
        # strip comments outside quotes
        buf, inq, i = [], False, 0
        while i < len(line):
            ch = line[i]
            if ch == '"':
                inq = not inq
            if not inq and (ch == "#" or line.startswith("//", i)):
                break
            buf.append(ch)
            i += 1
        s = "".join(buf)
        pos = 0
        while pos < len(s):
            m = _TOKEN.match(s, pos)
            if not m:
                if s[pos:].strip() == "":
                    break
                raise ValueError("cannot parse config near %r" % s[pos:pos + 30])
            pos = m.end()
            if m.group("brace"):
                out.append(("p", m.group("brace")))
            elif m.group("q") is not None:
                out.append(("s", m.group("q")))
            else:
                out.append(("w", m.group("w")))
        out.append(("nl", None))
    return out


def parse_string(text):
    toks = _tokens(text)
    pos = [0]

    def peek():
        while pos[0] < len(toks) and toks[pos[0]][0] == "nl":
            pos[0] += 1
        return toks[pos[0]] if pos[0] < len(toks) else None

    def value():
        t = peek()
        if t == ("p", "{"):
            pos[0] += 1
            return block(True)
        if t == ("p", "["):
            pos[0] += 1
            items = []
            while True:
                t = peek()
                if t is None:
                    raise ValueError("unterminated list")
                if t == ("p", "]"):
                    pos[0] += 1
                    return items
                if t == ("p", ","):
                    pos[0] += 1
                    continue
                items.append(value())
        pos[0] += 1
        if t[0] == "s":
            return t[1]
        # unquoted scalar; concatenate words up to the end of line ("a b c")
        words = [t[1]]
        while pos[0] < len(toks) and toks[pos[0]][0] == "w":
            words.append(toks[pos[0]][1])
            pos[0] += 1
        return _scalar(words[0]) if len(words) == 1 else " ".join(words)

    def put(tree, key, val):
        parts = key.split(".")
        for p in parts[:-1]:
            tree = tree.setdefault(p, Config())
        if isinstance(val, dict) and isinstance(tree.get(parts[-1]), dict):
            tree[parts[-1]].update(val)      # HOCON merges repeated blocks
        else:
            tree[parts[-1]] = val

    def block(nested):
        tree = Config()
        while True:
            t = peek()
            if t is None:
                if nested:
                    raise ValueError("unterminated block")
                return tree
            if t == ("p", "}"):
                pos[0] += 1
                if not nested:
                    raise ValueError("unexpected }")
                return tree
            if t == ("p", ","):
                pos[0] += 1
                continue
            if t[0] not in ("w", "s"):
                raise ValueError("expected a key, got %r" % (t[1],))
            key = t[1]
            pos[0] += 1
            t = peek()
            if t in (("p", "="), ("p", ":")):
                pos[0] += 1
                put(tree, key, value())
            elif t == ("p", "{"):
                pos[0] += 1
                put(tree, key, block(True))
            else:
                raise ValueError("expected = or { after key %r" % key)

    return block(False)


def parse_file(path):
    with open(path, "r") as f:
        return parse_string(f.read())


class ConfigFactory:
    """pyhocon.ConfigFactory stand-in (generate_ir_texture.py:36)"""
    parse_file = staticmethod(parse_file)
    parse_string = staticmethod(parse_string)
