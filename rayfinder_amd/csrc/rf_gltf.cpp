#include "rf_gltf.hpp"

#include "rf_bvh_gpu.hpp"
#include "rf_jpeg.hpp"

#include "rf_bvh.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <map>
#include <memory>
#include <stdexcept>
#include <string_view>

namespace fs = std::filesystem;

namespace rf
{
namespace
{
// ------------------------------------------------------------------------------------------------
// Minimal JSON document (enough for glTF).
// ------------------------------------------------------------------------------------------------
struct Json
{
    enum class Kind
    {
        Null,
        Bool,
        Number,
        String,
        Array,
        Object
    };
    Kind                                      kind = Kind::Null;
    bool                                      boolean = false;
    double                                    number = 0.0;
    std::string                               string;
    std::vector<Json>                         array;
    std::vector<std::pair<std::string, Json>> object;

    const Json* find(std::string_view key) const
    {
        for (const auto& kv : object)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Json& at(std::string_view key) const
    {
        const Json* j = find(key);
        if (!j) throw std::runtime_error("glTF: missing key '" + std::string(key) + "'");
        return *j;
    }
    bool        has(std::string_view key) const { return find(key) != nullptr; }
    std::size_t size() const { return kind == Kind::Array ? array.size() : object.size(); }
    // an index / count / offset / stride: a non-negative integer a double holds exactly (anything else in a file is hostile or
    // corrupt, and converting it would be undefined behaviour)
    std::size_t index() const
    {
        if (!(number >= 0.0) || number > 9007199254740992.0) throw std::runtime_error("glTF: a size or index is negative or beyond 2^53");
        return static_cast<std::size_t>(number);
    }
    float       f32() const { return static_cast<float>(number); } // cgltf: (float)atof(token)
};

class JsonParser
{
public:
    explicit JsonParser(std::string_view text) : mText(text) {}
    Json parse()
    {
        Json v = value();
        ws();
        return v;
    }

private:
    std::string_view mText;
    std::size_t      mPos = 0;

    [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("glTF JSON: ") + what); }
    void              ws()
    {
        while (mPos < mText.size() && (mText[mPos] == ' ' || mText[mPos] == '\n' || mText[mPos] == '\r' || mText[mPos] == '\t')) ++mPos;
    }
    char peek()
    {
        ws();
        if (mPos >= mText.size()) fail("unexpected end");
        return mText[mPos];
    }
    void expect(char c)
    {
        if (peek() != c) fail("unexpected character");
        ++mPos;
    }
    Json value()
    {
        const char c = peek();
        Json       v;
        if (c == '{')
        {
            v.kind = Json::Kind::Object;
            ++mPos;
            if (peek() == '}')
            {
                ++mPos;
                return v;
            }
            for (;;)
            {
                Json key = stringValue();
                expect(':');
                v.object.emplace_back(std::move(key.string), value());
                if (peek() == ',')
                {
                    ++mPos;
                    continue;
                }
                expect('}');
                break;
            }
        }
        else if (c == '[')
        {
            v.kind = Json::Kind::Array;
            ++mPos;
            if (peek() == ']')
            {
                ++mPos;
                return v;
            }
            for (;;)
            {
                v.array.push_back(value());
                if (peek() == ',')
                {
                    ++mPos;
                    continue;
                }
                expect(']');
                break;
            }
        }
        else if (c == '"')
        {
            v = stringValue();
        }
        else if (c == 't' || c == 'f')
        {
            v.kind = Json::Kind::Bool;
            v.boolean = c == 't';
            mPos += v.boolean ? 4 : 5;
        }
        else if (c == 'n')
        {
            mPos += 4;
        }
        else
        {
            v.kind = Json::Kind::Number;
            const std::string tmp(mText.substr(mPos, std::min<std::size_t>(64, mText.size() - mPos)));
            char*             end = nullptr;
            v.number = std::strtod(tmp.c_str(), &end);
            if (end == tmp.c_str()) fail("bad number");
            mPos += static_cast<std::size_t>(end - tmp.c_str());
        }
        return v;
    }
    Json stringValue()
    {
        expect('"');
        Json v;
        v.kind = Json::Kind::String;
        while (mPos < mText.size() && mText[mPos] != '"')
        {
            char c = mText[mPos++];
            if (c == '\\' && mPos < mText.size())
            {
                const char e = mText[mPos++];
                switch (e)
                {
                case 'n': c = '\n'; break;
                case 't': c = '\t'; break;
                case 'r': c = '\r'; break;
                case 'b': c = '\b'; break;
                case 'f': c = '\f'; break;
                case 'u':
                {
                    unsigned code = 0;
                    for (int i = 0; i < 4 && mPos < mText.size(); ++i) code = code * 16 + static_cast<unsigned>(std::strtol(std::string(1, mText[mPos++]).c_str(), nullptr, 16));
                    if (code < 0x80) c = static_cast<char>(code);
                    else
                    {
                        // UTF-8 encode (BMP only)
                        if (code < 0x800)
                        {
                            v.string.push_back(static_cast<char>(0xC0 | (code >> 6)));
                        }
                        else
                        {
                            v.string.push_back(static_cast<char>(0xE0 | (code >> 12)));
                            v.string.push_back(static_cast<char>(0x80 | ((code >> 6) & 0x3F)));
                        }
                        c = static_cast<char>(0x80 | (code & 0x3F));
                    }
                    break;
                }
                default: c = e; break;
                }
            }
            v.string.push_back(c);
        }
        if (mPos >= mText.size()) fail("unterminated string");
        ++mPos;
        return v;
    }
};

std::vector<uint8_t> readFile(const fs::path& path)
{
    FILE* fp = std::fopen(path.string().c_str(), "rb");
    if (!fp) throw std::runtime_error("Failed to open " + path.string() + ".");
    std::fseek(fp, 0, SEEK_END);
    const long n = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    std::vector<uint8_t> bytes(static_cast<std::size_t>(std::max<long>(n, 0)));
    if (!bytes.empty() && std::fread(bytes.data(), 1, bytes.size(), fp) != bytes.size())
    {
        std::fclose(fp);
        throw std::runtime_error("Short read from " + path.string() + ".");
    }
    std::fclose(fp);
    return bytes;
}

std::vector<uint8_t> decodeBase64(std::string_view s)
{
    std::vector<uint8_t> out;
    unsigned             acc = 0;
    int                  bits = 0;
    for (char c : s)
    {
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+') v = 62;
        else if (c == '/') v = 63;
        else continue;
        acc = (acc << 6) | static_cast<unsigned>(v);
        bits += 6;
        if (bits >= 8)
        {
            bits -= 8;
            out.push_back(static_cast<uint8_t>((acc >> bits) & 0xFF));
        }
    }
    return out;
}

std::string percentDecode(std::string_view s)
{
    std::string out;
    for (std::size_t i = 0; i < s.size(); ++i)
    {
        if (s[i] == '%' && i + 2 < s.size())
        {
            const std::string hex(s.substr(i + 1, 2));
            char*             end = nullptr;
            const long        v = std::strtol(hex.c_str(), &end, 16);
            if (end == hex.c_str() + 2)
            {
                out.push_back(static_cast<char>(v));
                i += 2;
                continue;
            }
        }
        out.push_back(s[i]);
    }
    return out;
}

struct Document
{
    Json                              json;
    std::vector<std::vector<uint8_t>> buffers;
    fs::path                          path;
};

std::vector<uint8_t> loadUri(const std::string& uri, const fs::path& gltfPath)
{
    if (uri.rfind("data:", 0) == 0)
    {
        const std::size_t comma = uri.find(',');
        if (comma == std::string::npos) throw std::runtime_error("glTF: malformed data URI");
        return decodeBase64(std::string_view(uri).substr(comma + 1));
    }
    return readFile(gltfPath.parent_path() / percentDecode(uri));
}

Document loadDocument(const fs::path& path)
{
    if (!fs::exists(path)) throw std::runtime_error("The gltf file " + path.string() + " does not exist.");
    const std::vector<uint8_t> bytes = readFile(path);
    Document                   doc;
    doc.path = path;
    std::vector<uint8_t> binChunk;
    bool                 haveBin = false;
    if (bytes.size() >= 12 && std::memcmp(bytes.data(), "glTF", 4) == 0)
    {
        uint32_t total;
        std::memcpy(&total, bytes.data() + 8, 4);
        std::size_t off = 12;
        bool        haveJson = false;
        while (off + 8 <= bytes.size() && off < total)
        {
            uint32_t len, type;
            std::memcpy(&len, bytes.data() + off, 4);
            std::memcpy(&type, bytes.data() + off + 4, 4);
            if (off + 8 + len > bytes.size()) throw std::runtime_error("Failed to parse gltf file " + path.string() + ".");
            if (type == 0x4E4F534Au && !haveJson)
            {
                doc.json = JsonParser(std::string_view(reinterpret_cast<const char*>(bytes.data() + off + 8), len)).parse();
                haveJson = true;
            }
            else if (type == 0x004E4942u && !haveBin)
            {
                binChunk.assign(bytes.begin() + static_cast<long>(off + 8), bytes.begin() + static_cast<long>(off + 8 + len));
                haveBin = true;
            }
            off += 8 + len;
        }
        if (!haveJson) throw std::runtime_error("Failed to parse gltf file " + path.string() + ".");
    }
    else
    {
        doc.json = JsonParser(std::string_view(reinterpret_cast<const char*>(bytes.data()), bytes.size())).parse();
    }
    if (const Json* buffers = doc.json.find("buffers"))
    {
        for (const Json& b : buffers->array)
        {
            if (const Json* uri = b.find("uri")) doc.buffers.push_back(loadUri(uri->string, path));
            else if (haveBin) doc.buffers.push_back(binChunk);
            else throw std::runtime_error("Failed to load gltf buffers for " + path.string() + ".");
        }
    }
    return doc;
}

// ------------------------------------------------------------------------------------------------
// glm 0.9.9.8 matrix arithmetic in f32 with the published operation order (column-major).
// ------------------------------------------------------------------------------------------------
struct Col4
{
    float x, y, z, w;
};
inline Col4 operator*(Col4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline Col4 operator+(Col4 a, Col4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline Col4 operator/(Col4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }

struct Mat4
{
    Col4 c[4];

    float at(int col, int row) const
    {
        const Col4& v = c[col];
        return row == 0 ? v.x : (row == 1 ? v.y : (row == 2 ? v.z : v.w));
    }
};

Mat4 identity() { return Mat4{{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; }

// operator*(mat4, mat4): Result[i] = ((A0*B[i][0] + A1*B[i][1]) + A2*B[i][2]) + A3*B[i][3]
Mat4 mul(const Mat4& a, const Mat4& b)
{
    Mat4 r;
    for (int i = 0; i < 4; ++i) r.c[i] = ((a.c[0] * b.c[i].x + a.c[1] * b.c[i].y) + a.c[2] * b.c[i].z) + a.c[3] * b.c[i].w;
    return r;
}

// operator*(mat4, vec4): (m0*v.x + m1*v.y) + (m2*v.z + m3*v.w)
Col4 mul(const Mat4& m, Col4 v) { return (m.c[0] * v.x + m.c[1] * v.y) + (m.c[2] * v.z + m.c[3] * v.w); }

Mat4 scaleMatrix(float x, float y, float z)
{
    const Mat4 m = identity();
    return Mat4{{m.c[0] * x, m.c[1] * y, m.c[2] * z, m.c[3]}};
}

Mat4 translateMatrix(float x, float y, float z)
{
    Mat4 m = identity();
    m.c[3] = ((m.c[0] * x + m.c[1] * y) + m.c[2] * z) + m.c[3];
    return m;
}

// glm::toMat4(quat) = mat4(mat3_cast(q)); quaternion stored x,y,z,w
Mat4 rotationMatrix(float x, float y, float z, float w)
{
    const float qxx = x * x, qyy = y * y, qzz = z * z;
    const float qxz = x * z, qxy = x * y, qyz = y * z;
    const float qwx = w * x, qwy = w * y, qwz = w * z;
    Mat4        r = identity();
    r.c[0] = {1.0f - 2.0f * (qyy + qzz), 2.0f * (qxy + qwz), 2.0f * (qxz - qwy), 0.0f};
    r.c[1] = {2.0f * (qxy - qwz), 1.0f - 2.0f * (qxx + qzz), 2.0f * (qyz + qwx), 0.0f};
    r.c[2] = {2.0f * (qxz + qwy), 2.0f * (qyz - qwx), 1.0f - 2.0f * (qxx + qyy), 0.0f};
    return r;
}

// glm::inverseTranspose(mat4) (gtc/matrix_inverse.inl): cofactors / determinant
Mat4 inverseTranspose(const Mat4& m)
{
    auto        e = [&m](int c, int r) { return m.at(c, r); };
    const float s00 = e(2, 2) * e(3, 3) - e(3, 2) * e(2, 3);
    const float s01 = e(2, 1) * e(3, 3) - e(3, 1) * e(2, 3);
    const float s02 = e(2, 1) * e(3, 2) - e(3, 1) * e(2, 2);
    const float s03 = e(2, 0) * e(3, 3) - e(3, 0) * e(2, 3);
    const float s04 = e(2, 0) * e(3, 2) - e(3, 0) * e(2, 2);
    const float s05 = e(2, 0) * e(3, 1) - e(3, 0) * e(2, 1);
    const float s06 = e(1, 2) * e(3, 3) - e(3, 2) * e(1, 3);
    const float s07 = e(1, 1) * e(3, 3) - e(3, 1) * e(1, 3);
    const float s08 = e(1, 1) * e(3, 2) - e(3, 1) * e(1, 2);
    const float s09 = e(1, 0) * e(3, 3) - e(3, 0) * e(1, 3);
    const float s10 = e(1, 0) * e(3, 2) - e(3, 0) * e(1, 2);
    const float s11 = e(1, 0) * e(3, 1) - e(3, 0) * e(1, 1);
    const float s12 = e(1, 2) * e(2, 3) - e(2, 2) * e(1, 3);
    const float s13 = e(1, 1) * e(2, 3) - e(2, 1) * e(1, 3);
    const float s14 = e(1, 1) * e(2, 2) - e(2, 1) * e(1, 2);
    const float s15 = e(1, 0) * e(2, 3) - e(2, 0) * e(1, 3);
    const float s16 = e(1, 0) * e(2, 2) - e(2, 0) * e(1, 2);
    const float s17 = e(1, 0) * e(2, 1) - e(2, 0) * e(1, 1);

    Mat4 inv;
    inv.c[0] = {+(e(1, 1) * s00 - e(1, 2) * s01 + e(1, 3) * s02), -(e(1, 0) * s00 - e(1, 2) * s03 + e(1, 3) * s04),
                +(e(1, 0) * s01 - e(1, 1) * s03 + e(1, 3) * s05), -(e(1, 0) * s02 - e(1, 1) * s04 + e(1, 2) * s05)};
    inv.c[1] = {-(e(0, 1) * s00 - e(0, 2) * s01 + e(0, 3) * s02), +(e(0, 0) * s00 - e(0, 2) * s03 + e(0, 3) * s04),
                -(e(0, 0) * s01 - e(0, 1) * s03 + e(0, 3) * s05), +(e(0, 0) * s02 - e(0, 1) * s04 + e(0, 2) * s05)};
    inv.c[2] = {+(e(0, 1) * s06 - e(0, 2) * s07 + e(0, 3) * s08), -(e(0, 0) * s06 - e(0, 2) * s09 + e(0, 3) * s10),
                +(e(0, 0) * s07 - e(0, 1) * s09 + e(0, 3) * s11), -(e(0, 0) * s08 - e(0, 1) * s10 + e(0, 2) * s11)};
    inv.c[3] = {-(e(0, 1) * s12 - e(0, 2) * s13 + e(0, 3) * s14), +(e(0, 0) * s12 - e(0, 2) * s15 + e(0, 3) * s16),
                -(e(0, 0) * s13 - e(0, 1) * s15 + e(0, 3) * s17), +(e(0, 0) * s14 - e(0, 1) * s16 + e(0, 2) * s17)};
    const float det = e(0, 0) * inv.c[0].x + e(0, 1) * inv.c[0].y + e(0, 2) * inv.c[0].z + e(0, 3) * inv.c[0].w;
    for (auto& col : inv.c) col = col / det;
    return inv;
}

struct MeshTransform
{
    Mat4 model = identity();
    Mat4 normal = identity();
};

Mat4 localMatrix(const Json& node)
{
    // fixed-size arrays of an untrusted file: checked, not assumed
    auto fixed = [](const Json* j, std::size_t n, const char* what) {
        if (j->array.size() != n) throw std::runtime_error(std::string("glTF: node.") + what + " must have " + std::to_string(n) + " elements");
    };
    if (const Json* m = node.find("matrix"))
    {
        fixed(m, 16, "matrix");
        Mat4 r;
        for (int c = 0; c < 4; ++c) r.c[c] = {m->array[4 * c].f32(), m->array[4 * c + 1].f32(), m->array[4 * c + 2].f32(), m->array[4 * c + 3].f32()};
        return r;
    }
    float s[3] = {1, 1, 1}, q[4] = {0, 0, 0, 1}, t[3] = {0, 0, 0};
    if (const Json* j = node.find("scale"))
    {
        fixed(j, 3, "scale");
        for (int i = 0; i < 3; ++i) s[i] = j->array[i].f32();
    }
    if (const Json* j = node.find("rotation"))
    {
        fixed(j, 4, "rotation");
        for (int i = 0; i < 4; ++i) q[i] = j->array[i].f32();
    }
    if (const Json* j = node.find("translation"))
    {
        fixed(j, 3, "translation");
        for (int i = 0; i < 3; ++i) t[i] = j->array[i].f32();
    }
    return mul(mul(translateMatrix(t[0], t[1], t[2]), rotationMatrix(q[0], q[1], q[2], q[3])), scaleMatrix(s[0], s[1], s[2]));
}

void walkNodes(const Json& nodes, std::size_t nodeIdx, const Mat4& parent, std::vector<MeshTransform>& out, int depth)
{
    if (depth > 256) throw std::runtime_error("glTF: node hierarchy too deep (cycle?)");
    const Json& node = nodes.array.at(nodeIdx);
    const Mat4  world = mul(parent, localMatrix(node));
    if (const Json* mesh = node.find("mesh"))
    {
        MeshTransform& t = out.at(mesh->index());
        t.model = world;
        t.normal = inverseTranspose(world);
    }
    if (const Json* children = node.find("children"))
        for (const Json& c : children->array) walkNodes(nodes, c.index(), world, out, depth + 1);
}

// ------------------------------------------------------------------------------------------------
// Accessors (the subset cgltf_accessor_unpack_floats / cgltf_accessor_read_uint cover)
// ------------------------------------------------------------------------------------------------
struct AccessorView
{
    const uint8_t* base = nullptr;
    std::size_t    count = 0, stride = 0;
    int            componentType = 0, components = 0;
    bool           normalized = false;
};

int componentSize(int type)
{
    switch (type)
    {
    case 5120:
    case 5121: return 1;
    case 5122:
    case 5123: return 2;
    case 5125:
    case 5126: return 4;
    default: throw std::runtime_error("glTF: unsupported accessor component type");
    }
}

struct BufferViewSpan
{
    const uint8_t* data = nullptr;
    std::size_t    size = 0, byteStride = 0;
};

// `count` elements of `elem` bytes, `stride` bytes apart, inside `avail` bytes -- without forming count * stride, which a hostile
// file can make wrap (byteStride and count are whatever the JSON says)
bool strideFits(std::size_t count, std::size_t stride, std::size_t elem, std::size_t avail)
{
    if (count == 0) return true;
    if (elem > avail) return false;
    if (count == 1) return true;
    return stride != 0 && count - 1 <= (avail - elem) / stride;
}

BufferViewSpan bufferViewSpan(const Document& doc, std::size_t index)
{
    const Json&                 bv = doc.json.at("bufferViews").array.at(index);
    const std::vector<uint8_t>& buffer = doc.buffers.at(bv.at("buffer").index());
    const std::size_t           offset = bv.has("byteOffset") ? bv.at("byteOffset").index() : 0;
    const std::size_t           length = bv.has("byteLength") ? bv.at("byteLength").index() : (offset <= buffer.size() ? buffer.size() - offset : 0);
    if (offset > buffer.size() || length > buffer.size() - offset) throw std::runtime_error("glTF: buffer view exceeds its buffer");
    BufferViewSpan s;
    s.data = buffer.data() + offset;
    s.size = length;
    s.byteStride = bv.has("byteStride") ? bv.at("byteStride").index() : 0;
    // 0 = "not given" = tightly packed.  glTF 2.0 asks for 4 .. 252 and a stride of at least the element; the reference loads through cgltf and never
    // calls cgltf_validate (gltf_model.cpp), so it reads a file that breaks those rules with the stride it states -- and so does this reader: what keeps a
    // hostile stride harmless is strideFits() below (the last element must end inside the view), not a range check
    return s;
}

// base == nullptr: the accessor has no buffer view (cgltf then reads zeros; only meaningful together with "sparse")
AccessorView accessorView(const Document& doc, std::size_t index)
{
    const Json&  acc = doc.json.at("accessors").array.at(index);
    AccessorView v;
    v.componentType = static_cast<int>(acc.at("componentType").number);
    const std::string& type = acc.at("type").string;
    v.components = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT4" ? 16 : 0;
    if (v.components == 0) throw std::runtime_error("glTF: unsupported accessor type " + type);
    v.count = acc.at("count").index();
    v.normalized = acc.has("normalized") && acc.at("normalized").boolean;
    const std::size_t elem = static_cast<std::size_t>(componentSize(v.componentType)) * static_cast<std::size_t>(v.components);
    v.stride = elem;
    if (!acc.has("bufferView")) return v;
    const BufferViewSpan bv = bufferViewSpan(doc, acc.at("bufferView").index());
    if (bv.byteStride != 0) v.stride = bv.byteStride; // (smaller than the element: overlapping reads, as cgltf does them)
    const std::size_t offset = acc.has("byteOffset") ? acc.at("byteOffset").index() : 0;
    if (v.count && (offset > bv.size || !strideFits(v.count, v.stride, elem, bv.size - offset))) throw std::runtime_error("glTF: accessor exceeds its buffer view");
    v.base = bv.data + offset;
    return v;
}

// cgltf 1.13 cgltf_component_read_float (un-vendored dependency, published algorithm): f32 as is; normalized integers divided by
// 127 / 255 / 32767 / 65535 WITHOUT the clamp to -1 the glTF text suggests for the most negative value (so -128 -> -1.00787...);
// anything else converted as an integer
float componentAsFloat(const uint8_t* p, int type, bool normalized)
{
    switch (type)
    {
    case 5126:
    {
        float f;
        std::memcpy(&f, p, 4);
        return f;
    }
    case 5120:
    {
        const int8_t v = static_cast<int8_t>(*p);
        return normalized ? static_cast<float>(v) / 127.0f : static_cast<float>(v);
    }
    case 5121: return normalized ? static_cast<float>(*p) / 255.0f : static_cast<float>(*p);
    case 5122:
    {
        int16_t v;
        std::memcpy(&v, p, 2);
        return normalized ? static_cast<float>(v) / 32767.0f : static_cast<float>(v);
    }
    case 5123:
    {
        uint16_t v;
        std::memcpy(&v, p, 2);
        return normalized ? static_cast<float>(v) / 65535.0f : static_cast<float>(v);
    }
    default:
    {
        uint32_t v;
        std::memcpy(&v, p, 4);
        return static_cast<float>(v);
    }
    }
}

uint32_t componentAsUint(const uint8_t* p, int type)
{
    switch (type)
    {
    case 5121: return *p;
    case 5123:
    {
        uint16_t v;
        std::memcpy(&v, p, 2);
        return v;
    }
    case 5125:
    {
        uint32_t v;
        std::memcpy(&v, p, 4);
        return v;
    }
    default: throw std::runtime_error("glTF: index accessor must be unsigned");
    }
}

// cgltf_accessor_unpack_floats (gltf_model.cpp:400-438 reads POSITION / NORMAL / TEXCOORD_0 through it; cgltf 1.13, published
// algorithm): first the base accessor -- zeros when it has no buffer view, any component type, normalized integers as above,
// the buffer view's byteStride -- then the "sparse" overlay: value k (laid out with the ACCESSOR's stride, which without a
// buffer view is the element size) replaces element indices[k].
std::vector<float> unpackFloats(const Document& doc, std::size_t accessorIndex, int comps, const char* what)
{
    const AccessorView v = accessorView(doc, accessorIndex);
    if (v.components != comps) throw std::runtime_error(std::string("glTF: ") + what + " has the wrong accessor type");
    // (an accessor over a buffer view is bounded by that view; one with sparse data only is bounded by nothing but this)
    if (v.count > (std::size_t(1) << 32)) throw std::runtime_error(std::string("glTF: ") + what + " accessor count is implausible");
    std::vector<float> out(v.count * static_cast<std::size_t>(comps), 0.0f);
    const std::size_t  cs = static_cast<std::size_t>(componentSize(v.componentType));
    if (v.base != nullptr)
        for (std::size_t i = 0; i < v.count; ++i)
            for (int c = 0; c < comps; ++c)
                out[i * comps + c] = componentAsFloat(v.base + i * v.stride + static_cast<std::size_t>(c) * cs, v.componentType, v.normalized);
    const Json& acc = doc.json.at("accessors").array.at(accessorIndex);
    if (acc.has("sparse"))
    {
        const Json&       sp = acc.at("sparse");
        const std::size_t n = sp.at("count").index();
        const Json&       si = sp.at("indices");
        const Json&       sv = sp.at("values");
        const int         indexType = static_cast<int>(si.at("componentType").number);
        if (indexType != 5121 && indexType != 5123 && indexType != 5125) throw std::runtime_error("glTF: sparse indices must be unsigned");
        const BufferViewSpan ib = bufferViewSpan(doc, si.at("bufferView").index()), vb = bufferViewSpan(doc, sv.at("bufferView").index());
        const std::size_t    io = si.has("byteOffset") ? si.at("byteOffset").index() : 0, vo = sv.has("byteOffset") ? sv.at("byteOffset").index() : 0;
        const std::size_t    is = static_cast<std::size_t>(componentSize(indexType)), elem = cs * static_cast<std::size_t>(comps);
        if (n && (io > ib.size || !strideFits(n, is, is, ib.size - io))) throw std::runtime_error("glTF: sparse indices exceed their buffer view");
        if (n && (vo > vb.size || !strideFits(n, v.stride, elem, vb.size - vo))) throw std::runtime_error("glTF: sparse values exceed their buffer view");
        for (std::size_t k = 0; k < n; ++k)
        {
            const std::size_t at = componentAsUint(ib.data + io + k * is, indexType);
            if (at >= v.count) throw std::runtime_error("glTF: sparse index out of range");
            for (int c = 0; c < comps; ++c)
                out[at * comps + c] = componentAsFloat(vb.data + vo + k * v.stride + static_cast<std::size_t>(c) * cs, v.componentType, v.normalized);
        }
    }
    else if (v.base == nullptr) throw std::runtime_error(std::string("glTF: ") + what + " accessor has neither a buffer view nor sparse data");
    return out;
}

// ------------------------------------------------------------------------------------------------
// PNG -> RGBA8, following what stb_image returns for req_comp = 4 (texture.cpp:12-31)
// ------------------------------------------------------------------------------------------------
uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

int paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

// Un-filter one (sub)image of w x h pixels with `bpp` bytes per complete pixel (min 1) and
// `rowBytes` bytes per row; `in` has h rows of 1 + rowBytes bytes.
void unfilter(const uint8_t* in, uint8_t* out, std::size_t rowBytes, std::size_t h, std::size_t bpp)
{
    std::vector<uint8_t> zero(rowBytes, 0);
    for (std::size_t y = 0; y < h; ++y)
    {
        const uint8_t  filter = in[y * (rowBytes + 1)];
        const uint8_t* src = in + y * (rowBytes + 1) + 1;
        uint8_t*       dst = out + y * rowBytes;
        const uint8_t* up = y ? out + (y - 1) * rowBytes : zero.data();
        for (std::size_t i = 0; i < rowBytes; ++i)
        {
            const int a = i >= bpp ? dst[i - bpp] : 0;
            const int b = up[i];
            const int c = i >= bpp ? up[i - bpp] : 0;
            int       v = src[i];
            switch (filter)
            {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: throw std::runtime_error("PNG: invalid filter");
            }
            dst[i] = static_cast<uint8_t>(v);
        }
    }
}

Rgba8Image decodePng(std::span<const uint8_t> data)
{
    static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (data.size() < 8 || std::memcmp(data.data(), kSig, 8) != 0) throw std::runtime_error("image is neither PNG nor JPEG (the only texture containers supported)");
    uint32_t             width = 0, height = 0;
    int                  depth = 0, colorType = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    std::size_t          off = 8;
    bool                 end = false;
    while (!end && off + 12 <= data.size())
    {
        const uint32_t len = be32(data.data() + off);
        const uint8_t* type = data.data() + off + 4;
        const uint8_t* body = data.data() + off + 8;
        if (off + 12 + len > data.size()) throw std::runtime_error("PNG: truncated chunk");
        if (!std::memcmp(type, "IHDR", 4))
        {
            if (len != 13) throw std::runtime_error("PNG: IHDR chunk is not 13 bytes long");
            width = be32(body);
            height = be32(body + 4);
            depth = body[8];
            colorType = body[9];
            interlace = body[12];
        }
        else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        off += 12 + len;
    }
    if (width == 0 || height == 0) throw std::runtime_error("PNG: missing IHDR");
    const int channels = colorType == 0 ? 1 : colorType == 2 ? 3 : colorType == 3 ? 1 : colorType == 4 ? 2 : colorType == 6 ? 4 : 0;
    if (channels == 0) throw std::runtime_error("PNG: bad colour type");
    // PNG spec table 11.1: grey 1/2/4/8/16, truecolour and the alpha types 8/16, palette 1/2/4/8
    const bool depthOk = colorType == 0   ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                         : colorType == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                          : (depth == 8 || depth == 16);
    if (!depthOk) throw std::runtime_error("PNG: bit depth " + std::to_string(depth) + " is not allowed for colour type " + std::to_string(colorType));
    if (interlace > 1) throw std::runtime_error("PNG: unknown interlace method");

    const std::size_t bitsPerPixel = static_cast<std::size_t>(channels) * static_cast<std::size_t>(depth);
    const std::size_t bpp = std::max<std::size_t>(1, bitsPerPixel / 8);

    // inflate everything
    std::size_t rawSize = 0;
    struct Pass
    {
        uint32_t x0, y0, dx, dy, w, h;
    };
    std::vector<Pass> passes;
    if (!interlace) passes.push_back({0, 0, 1, 1, width, height});
    else
    {
        static const uint32_t xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dxs[7] = {8, 8, 4, 4, 2, 2, 1}, dys[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int p = 0; p < 7; ++p)
        {
            const uint32_t w = (width - xs[p] + dxs[p] - 1) / dxs[p], h = (height - ys[p] + dys[p] - 1) / dys[p];
            if (width > xs[p] && height > ys[p] && w && h) passes.push_back({xs[p], ys[p], dxs[p], dys[p], w, h});
        }
    }
    for (const Pass& p : passes) rawSize += (1 + (p.w * bitsPerPixel + 7) / 8) * p.h;
    std::vector<uint8_t> raw(rawSize);
    {
        uLongf    destLen = static_cast<uLongf>(raw.size());
        const int rc = uncompress(raw.data(), &destLen, idat.data(), static_cast<uLong>(idat.size()));
        if (rc != Z_OK || destLen != raw.size()) throw std::runtime_error("PNG: inflate failed");
    }

    Rgba8Image img;
    img.width = width;
    img.height = height;
    img.rgba.assign(static_cast<std::size_t>(width) * height * 4, 255);
    // stb scales low-bit-depth grey to 8 bits by these factors; palette indices are not scaled
    static const int kDepthScale[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};

    std::size_t rawOff = 0;
    for (const Pass& p : passes)
    {
        const std::size_t    rowBytes = (p.w * bitsPerPixel + 7) / 8;
        std::vector<uint8_t> rows(rowBytes * p.h);
        unfilter(raw.data() + rawOff, rows.data(), rowBytes, p.h, bpp);
        rawOff += (1 + rowBytes) * p.h;
        for (uint32_t y = 0; y < p.h; ++y)
        {
            const uint8_t* row = rows.data() + y * rowBytes;
            for (uint32_t x = 0; x < p.w; ++x)
            {
                int s[4] = {0, 0, 0, 255};
                for (int c = 0; c < channels; ++c)
                {
                    const std::size_t sampleIdx = static_cast<std::size_t>(x) * channels + c;
                    int               v;
                    if (depth == 8) v = row[sampleIdx];
                    else if (depth == 16) v = row[2 * sampleIdx]; // stb keeps the high byte
                    else
                    {
                        const std::size_t bit = sampleIdx * depth;
                        v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                        if (colorType == 0) v *= kDepthScale[depth];
                    }
                    s[c] = v;
                }
                uint8_t* px = img.rgba.data() + 4 * (static_cast<std::size_t>(p.y0 + y * p.dy) * width + (p.x0 + x * p.dx));
                switch (colorType)
                {
                case 0: px[0] = px[1] = px[2] = static_cast<uint8_t>(s[0]); break;
                case 2:
                    px[0] = static_cast<uint8_t>(s[0]);
                    px[1] = static_cast<uint8_t>(s[1]);
                    px[2] = static_cast<uint8_t>(s[2]);
                    break;
                case 3:
                {
                    const std::size_t idx = static_cast<std::size_t>(s[0]);
                    if (3 * idx + 2 >= palette.size()) throw std::runtime_error("PNG: palette index out of range");
                    px[0] = palette[3 * idx];
                    px[1] = palette[3 * idx + 1];
                    px[2] = palette[3 * idx + 2];
                    px[3] = idx < trns.size() ? trns[idx] : 255;
                    break;
                }
                case 4:
                    px[0] = px[1] = px[2] = static_cast<uint8_t>(s[0]);
                    px[3] = static_cast<uint8_t>(s[1]);
                    break;
                default:
                    px[0] = static_cast<uint8_t>(s[0]);
                    px[1] = static_cast<uint8_t>(s[1]);
                    px[2] = static_cast<uint8_t>(s[2]);
                    px[3] = static_cast<uint8_t>(s[3]);
                    break;
                }
            }
        }
    }
    return img;
}

// FNV-1a over the four base colour floats, used to de-duplicate factor-only materials
uint32_t hashFactor(const float (&f)[4])
{
    uint32_t       h = 2166136261u;
    const uint8_t* p = reinterpret_cast<const uint8_t*>(f);
    for (std::size_t i = 0; i < sizeof f; ++i)
    {
        h ^= p[i];
        h *= 16777619u;
    }
    return h;
}
} // namespace

Texture textureFromMemory(std::span<const uint8_t> data)
{
    // stb_image sniffs the container (texture.cpp:12-31 passes whatever the glTF embeds): PNG or JPEG
    const Rgba8Image img = looksLikeJpeg(data) ? decodeJpeg(data) : decodePng(data);
    Texture          t;
    t.width = img.width;
    t.height = img.height;
    t.pixels.resize(static_cast<std::size_t>(img.width) * img.height);
    for (std::size_t i = 0; i < t.pixels.size(); ++i)
    {
        const uint32_t r = img.rgba[4 * i], g = img.rgba[4 * i + 1], b = img.rgba[4 * i + 2];
        t.pixels[i] = b | (g << 8) | (r << 16) | (255u << 24);
    }
    return t;
}

Texture textureFromPixel(float r, float g, float b, float a)
{
    const uint32_t r8 = static_cast<uint32_t>(r * 255.0f), g8 = static_cast<uint32_t>(g * 255.0f);
    const uint32_t b8 = static_cast<uint32_t>(b * 255.0f), a8 = static_cast<uint32_t>(a * 255.0f);
    Texture        t;
    t.width = t.height = 1;
    t.pixels = {b8 | (g8 << 8) | (r8 << 16) | (a8 << 24)};
    return t;
}

GltfModel loadGltfModel(const std::string& pathString)
{
    const Document doc = loadDocument(pathString);
    const Json&    js = doc.json;
    const Json&    meshes = js.at("meshes");

    std::vector<MeshTransform> transforms(meshes.size());
    {
        const Json& scenes = js.at("scenes");
        if (scenes.size() != 1) throw std::runtime_error("glTF: exactly one scene is supported"); // gltf_model.cpp:296
        const std::size_t sceneIdx = js.has("scene") ? js.at("scene").index() : 0;
        const Json&       scene = scenes.array.at(sceneIdx);
        if (const Json* roots = scene.find("nodes"))
            for (const Json& n : roots->array) walkNodes(js.at("nodes"), n.index(), identity(), transforms, 0);
    }

    GltfModel                                model;
    std::vector<std::pair<std::size_t, std::size_t>> imageLookup;  // gltf image -> texture
    std::vector<std::pair<uint32_t, std::size_t>>    factorLookup; // hash -> texture

    for (std::size_t meshIdx = 0; meshIdx < meshes.size(); ++meshIdx)
    {
        const MeshTransform& xf = transforms[meshIdx];
        for (const Json& prim : meshes.array[meshIdx].at("primitives").array)
        {
            if (prim.has("mode") && prim.at("mode").index() != 4) throw std::runtime_error("glTF: only triangle primitives are supported");
            if (!prim.has("material")) throw std::runtime_error("glTF: primitive without material");
            const Json& material = js.at("materials").array.at(prim.at("material").index());
            static const Json kEmpty = [] {
                Json j;
                j.kind = Json::Kind::Object;
                return j;
            }();
            const Json& pbr = material.has("pbrMetallicRoughness") ? material.at("pbrMetallicRoughness") : kEmpty;

            GltfMesh mesh;
            if (const Json* bct = pbr.find("baseColorTexture"))
            {
                if (bct->has("texCoord") && bct->at("texCoord").index() != 0) throw std::runtime_error("glTF: base colour texture must use TEXCOORD_0");
                const Json& texture = js.at("textures").array.at(bct->at("index").index());
                if (texture.has("sampler"))
                {
                    const Json& sampler = js.at("samplers").array.at(texture.at("sampler").index());
                    const auto  wrap = [&](const char* k) { return sampler.has(k) ? sampler.at(k).index() : 10497u; };
                    if (wrap("wrapS") != 10497u || wrap("wrapT") != 10497u) throw std::runtime_error("glTF: only REPEAT texture wrapping is supported");
                }
                const std::size_t imageIdx = texture.at("source").index();
                auto found = std::find_if(imageLookup.begin(), imageLookup.end(), [&](const auto& kv) { return kv.first == imageIdx; });
                if (found == imageLookup.end())
                {
                    const Json&          image = js.at("images").array.at(imageIdx);
                    std::vector<uint8_t> bytes;
                    if (const Json* bvIdx = image.find("bufferView"))
                    {
                        const Json&                 bv = js.at("bufferViews").array.at(bvIdx->index());
                        const std::vector<uint8_t>& buf = doc.buffers.at(bv.at("buffer").index());
                        const std::size_t           o = bv.has("byteOffset") ? bv.at("byteOffset").index() : 0, n = bv.at("byteLength").index();
                        if (o + n > buf.size()) throw std::runtime_error("glTF: image buffer view exceeds its buffer");
                        bytes.assign(buf.begin() + static_cast<long>(o), buf.begin() + static_cast<long>(o + n));
                    }
                    else
                    {
                        const std::string uri = image.at("uri").string;
                        if (uri.rfind("data:", 0) != 0)
                        {
                            const fs::path p = doc.path.parent_path() / percentDecode(uri);
                            if (!fs::exists(p)) throw std::runtime_error("The image " + p.string() + " does not exist.");
                        }
                        bytes = loadUri(uri, doc.path);
                    }
                    mesh.baseColorTextureIndex = model.baseColorTextures.size();
                    imageLookup.emplace_back(imageIdx, mesh.baseColorTextureIndex);
                    model.baseColorTextures.push_back(textureFromMemory(bytes));
                }
                else
                {
                    mesh.baseColorTextureIndex = found->second;
                }
            }
            else
            {
                float factor[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (const Json* f = pbr.find("baseColorFactor"))
                    for (int i = 0; i < 4; ++i) factor[i] = f->array.at(static_cast<std::size_t>(i)).f32();
                const uint32_t h = hashFactor(factor);
                auto found = std::find_if(factorLookup.begin(), factorLookup.end(), [&](const auto& kv) { return kv.first == h; });
                if (found == factorLookup.end())
                {
                    mesh.baseColorTextureIndex = model.baseColorTextures.size();
                    factorLookup.emplace_back(h, mesh.baseColorTextureIndex);
                    model.baseColorTextures.push_back(textureFromPixel(factor[0], factor[1], factor[2], factor[3]));
                }
                else
                {
                    mesh.baseColorTextureIndex = found->second;
                }
            }

            // indices (cgltf_accessor_read_uint)
            {
                if (!prim.has("indices")) throw std::runtime_error("glTF: non-indexed primitives are not supported");
                const AccessorView v = accessorView(doc, prim.at("indices").index());
                if (v.components != 1) throw std::runtime_error("glTF: index accessor must be SCALAR");
                // (cgltf_accessor_read_uint returns 0 for a sparse accessor and for one without a buffer view: the reference asserts)
                if (v.base == nullptr || doc.json.at("accessors").array.at(prim.at("indices").index()).has("sparse")) throw std::runtime_error("glTF: sparse index accessors are not supported");
                if (v.count % 3 != 0) throw std::runtime_error("glTF: index count is not a multiple of 3");
                mesh.indices.resize(v.count);
                for (std::size_t i = 0; i < v.count; ++i) mesh.indices[i] = componentAsUint(v.base + i * v.stride, v.componentType);
            }
            // attributes (cgltf_accessor_unpack_floats)
            const Json& attrs = prim.at("attributes");
            const auto  floats = [&](const char* name, int comps) {
                if (!attrs.has(name)) throw std::runtime_error(std::string("glTF: primitive lacks ") + name);
                return unpackFloats(doc, attrs.at(name).index(), comps, name);
            };
            const std::vector<float> pos = floats("POSITION", 3), nrm = floats("NORMAL", 3), uv = floats("TEXCOORD_0", 2);
            if (pos.size() != nrm.size() || pos.size() / 3 != uv.size() / 2) throw std::runtime_error("glTF: attribute counts differ");
            const std::size_t vertexCount = pos.size() / 3;
            mesh.positions.resize(vertexCount);
            mesh.normals.resize(vertexCount);
            mesh.texCoords.resize(vertexCount);
            for (std::size_t i = 0; i < vertexCount; ++i)
            {
                const Col4 p = mul(xf.model, Col4{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], 1.0f}); // gltf_model.cpp:413
                mesh.positions[i] = vec3(p.x, p.y, p.z);
                // glm::normalize on the vec4 product, then truncated to vec3 (gltf_model.cpp:428)
                const Col4  n = mul(xf.normal, Col4{nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2], 0.0f});
                const float d = (n.x * n.x + n.y * n.y) + (n.z * n.z + n.w * n.w);
                const float inv = 1.0f / std::sqrt(d);
                mesh.normals[i] = vec3(n.x * inv, n.y * inv, n.z * inv);
                mesh.texCoords[i] = Vec2{uv[2 * i], uv[2 * i + 1]};
            }
            for (uint32_t idx : mesh.indices)
                if (idx >= vertexCount) throw std::runtime_error("glTF: vertex index out of range");
            model.meshes.push_back(std::move(mesh));
        }
    }
    std::sort(model.meshes.begin(), model.meshes.end(),
              [](const GltfMesh& a, const GltfMesh& b) { return a.baseColorTextureIndex < b.baseColorTextureIndex; });
    return model;
}

namespace
{
// which builder the bake uses: -1 = host (the reference's recursion restated, rf_bvh.cpp), >= 0 = that GPU (rf_bvh_gpu.hip)
int gBakeBvhDevice = -1;
} // namespace

void setBakeBvhBuilder(int gpuDeviceOrMinusOne) { gBakeBvhDevice = gpuDeviceOrMinusOne; }

PtFormat ptFormatFromTriangles(std::span<const Positions> positions, std::span<const Normals> normals, std::span<const TexCoords> texCoords,
                               std::span<const uint32_t> textureIndices, std::vector<Texture> textures)
{
    PtFormat  out;
    const Bvh bvh = gBakeBvhDevice >= 0 ? buildBvhGpu(positions, gBakeBvhDevice) : buildBvh(positions);
    const std::span<const std::size_t> order(bvh.triangleIndices);
    out.bvhNodes = bvh.nodes;
    out.bvhPositionAttributes = reorderAttributes(positions, order);
    const std::vector<Normals>   n = reorderAttributes(normals, order);
    const std::vector<TexCoords> t = reorderAttributes(texCoords, order);
    const std::vector<uint32_t>  ti = reorderAttributes(textureIndices, order);
    out.trianglePositionAttributes.reserve(positions.size());
    out.triangleVertexAttributes.reserve(positions.size());
    for (std::size_t i = 0; i < positions.size(); ++i)
    {
        const Positions& p = out.bvhPositionAttributes[i];
        out.trianglePositionAttributes.push_back(PositionAttribute{p.v0, 0.0f, p.v1, 0.0f, p.v2, 0.0f});
        out.triangleVertexAttributes.push_back(VertexAttributes{n[i].n0, 0.0f, n[i].n1, 0.0f, n[i].n2, 0.0f, t[i].uv0, t[i].uv1, t[i].uv2, ti[i], 0u});
    }
    out.baseColorTextures = std::move(textures);
    return out;
}

PtFormat ptFormatFromGltf(const std::string& path)
{
    GltfModel model = loadGltfModel(path);

    // un-index (flattened_model.cpp:22-43)
    std::vector<Positions> positions;
    std::vector<Normals>   normals;
    std::vector<TexCoords> texCoords;
    std::vector<uint32_t>  textureIndices;
    for (const GltfMesh& m : model.meshes)
    {
        for (std::size_t i = 0; i + 2 < m.indices.size(); i += 3)
        {
            const uint32_t a = m.indices[i], b = m.indices[i + 1], c = m.indices[i + 2];
            positions.push_back({m.positions[a], m.positions[b], m.positions[c]});
            normals.push_back({m.normals[a], m.normals[b], m.normals[c]});
            texCoords.push_back({m.texCoords[a], m.texCoords[b], m.texCoords[c]});
            textureIndices.push_back(static_cast<uint32_t>(m.baseColorTextureIndex));
        }
    }
    if (positions.empty()) throw std::runtime_error("glTF: model has no triangles");

    PtFormat out = ptFormatFromTriangles(positions, normals, texCoords, textureIndices, {});

    // raster-mesh arrays (pt_format.cpp:85-148)
    for (const GltfMesh& m : model.meshes)
    {
        const uint64_t vertexOffset = out.vertexPositions.size(), numVertices = m.positions.size();
        for (const Vec3& p : m.positions) out.vertexPositions.push_back({p.x, p.y, p.z, 1.0f});
        for (const Vec3& n : m.normals) out.vertexNormals.push_back({n.x, n.y, n.z, 0.0f});
        out.vertexTexCoords.insert(out.vertexTexCoords.end(), m.texCoords.begin(), m.texCoords.end());
        out.modelVertexPositions.push_back({vertexOffset, numVertices});
        out.modelVertexNormals.push_back({vertexOffset, numVertices});
        out.modelVertexTexCoords.push_back({vertexOffset, numVertices});
        const uint64_t indexOffset = out.vertexIndices.size();
        out.vertexIndices.insert(out.vertexIndices.end(), m.indices.begin(), m.indices.end());
        out.modelVertexIndices.push_back({indexOffset, m.indices.size()});
        out.modelBaseColorTextureIndices.push_back(static_cast<uint32_t>(m.baseColorTextureIndex));
    }
    out.baseColorTextures = std::move(model.baseColorTextures);
    return out;
}
} // namespace rf
